// marex_hip.hip -- hand-written gfx950 (CDNA4) kernels + C ABI for the preprocess_data hot path.
//
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -shared (see csrc/build.py).
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
    unsigned char* thr_scratch = nullptr;  // device, per-(tile, day, lane) state bytes of the 1024-thread threshold tiles
    size_t thr_scratch_bytes = 0;
    unsigned char* detrend_scratch = nullptr;  // device, partial sums / coefficients / means of the detrend reductions
    size_t detrend_scratch_bytes = 0;
};

static int fail(marex_ctx* ctx, int code, const char* fmt, ...) {
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

static void drain_timers(marex_ctx* ctx) {
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

static int env_int(const char* name, int dflt) {
    const char* s = getenv(name);
    return (s && *s) ? atoi(s) : dflt;
}

extern "C" int marex_abi_version(void) { return MAREX_ABI_VERSION; }

extern "C" int marex_create(int device, marex_ctx** out) {
    if (!out) return -1;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || device < 0 || device >= n) return -3;
    marex_ctx* c = new marex_ctx();
    c->device = device;
    *out = c;
    return 0;
}

extern "C" int marex_destroy(marex_ctx* ctx) {
    if (!ctx) return -1;
    drain_timers(ctx);
    if (ctx->shift_info) (void)hipFree(ctx->shift_info);
    if (ctx->thr_scratch) (void)hipFree(ctx->thr_scratch);
    if (ctx->detrend_scratch) (void)hipFree(ctx->detrend_scratch);
    delete ctx;
    return 0;
}

extern "C" const char* marex_last_error(marex_ctx* ctx) { return ctx ? ctx->err.c_str() : "null context"; }

extern "C" int marex_set_stream(marex_ctx* ctx, void* s) {
    if (!ctx) return -1;
    ctx->stream = (hipStream_t)s;
    return 0;
}

extern "C" int marex_sync(marex_ctx* ctx) {
    if (!ctx) return -1;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    drain_timers(ctx);
    return 0;
}

extern "C" int marex_timing_enable(marex_ctx* ctx, int on) {
    if (!ctx) return -1;
    ctx->timing = on != 0;
    return 0;
}

extern "C" int marex_timing_reset(marex_ctx* ctx) {
    if (!ctx) return -1;
    drain_timers(ctx);
    memset(ctx->total_ms, 0, sizeof ctx->total_ms);
    memset(ctx->launches, 0, sizeof ctx->launches);
    return 0;
}

extern "C" int marex_timing_get(marex_ctx* ctx, int kid, double* total_ms, int64_t* launches) {
    if (!ctx || kid < 0 || kid >= MAREX_K_COUNT) return -1;
    drain_timers(ctx);
    if (total_ms) *total_ms = ctx->total_ms[kid];
    if (launches) *launches = ctx->launches[kid];
    return 0;
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
    int k = 1 + (int)((v - e[1]) * inv_width);
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
    int k = 1 + (int)((v - first) * inv_width);
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

// ------------------------------------------------------------------------------------------------
// synthetic field (marex_amd/synth.py)
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned long long mix64(unsigned long long z) {
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
__device__ __forceinline__ unsigned sum16(unsigned long long h) {
    return (unsigned)(h & 0xFFFF) + (unsigned)((h >> 16) & 0xFFFF) + (unsigned)((h >> 32) & 0xFFFF) + (unsigned)(h >> 48);
}

__global__ void __launch_bounds__(256) k_synth(const float* __restrict__ mean, const float* __restrict__ amp,
                                               const unsigned char* __restrict__ hemi,
                                               const unsigned char* __restrict__ land,
                                               const float* __restrict__ seas, const float* __restrict__ trend,
                                               unsigned long long seed, long cell_base, long T, long C,
                                               float z_scale, float noise_amp, int rows_per_block,
                                               float* __restrict__ x) {
    long c = (long)blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    long t0 = (long)blockIdx.y * rows_per_block;
    long t1 = t0 + rows_per_block < T ? t0 + rows_per_block : T;
    const float m = mean[c], a = amp[c];
    const int hm = hemi[c];
    const bool is_land = land[c] != 0;
    const unsigned long long kc = seed * 0x9E3779B97F4A7C15ull + (unsigned long long)(c + cell_base) * 0x8CB92BA72F3D8DD7ull;
    for (long t = t0; t < t1; ++t) {
        float v;
        if (is_land) {
            v = nan_f();
        } else {
            unsigned long long key = kc + (unsigned long long)t * 0xD1B54A32D192ED03ull;
            unsigned s = sum16(mix64(key)) + sum16(mix64(key ^ 0xA5A5A5A5A5A5A5A5ull));
            float z = (float)(2 * (int)s - 8 * 65535) * z_scale;
            float sa = a * seas[2 * t + hm];
            float b = m + sa;
            float cc = b + trend[t];
            v = cc + noise_amp * z;
        }
        x[(size_t)t * C + c] = v;
    }
}

extern "C" int marex_synth_sst_f32(marex_ctx* ctx, const float* mean, const float* amp, const uint8_t* hemi,
                                   const uint8_t* land, const float* seas, const float* trend, uint64_t seed,
                                   int64_t cell_base, int64_t T, int64_t C, float* x) {
    if (!ctx) return -1;
    if (!mean || !amp || !hemi || !land || !seas || !trend || !x || T <= 0 || C <= 0)
        return fail(ctx, -1, "marex_synth_sst_f32: bad argument");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const int rows = 64;
    dim3 grid((unsigned)((C + 255) / 256), (unsigned)((T + rows - 1) / rows));
    const float z_scale = (float)(1.0 / (2.0 * sqrt(8.0 * (65536.0 * 65536.0 - 1.0) / 12.0)));
    {
        LaunchTimer lt(ctx, MAREX_K_SYNTH);
        hipLaunchKernelGGL(k_synth, grid, dim3(256), 0, ctx->stream, mean, amp, hemi, land, seas, trend,
                           (unsigned long long)seed, (long)cell_base, (long)T, (long)C, z_scale, 0.8f, rows, x);
    }
    HIP_TRY(ctx, hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------------------------------------
// K_A: shifting-baseline anomaly  (smoothing + rolling climatology + anomaly + bins + validation)
//
// Work item = (block of 256 consecutive cells, chunk of D consecutive dayofyears).  The workgroup
// walks the calendar years in ascending order.  For one year the D dayofyears of the chunk are D
// consecutive timesteps, so ONE load of D+S-1 rows (256 contiguous floats each) feeds the S-step
// smoothing of all D days out of registers.  The W-year history of every (cell, dayofyear) lives in
// an LDS ring [D][W][256] that is private per lane (no barriers in the year loop).  Each input row
// is read by ceil((D+S-1)/D) chunks; blocks of one cell block are placed on one XCD so that those
// re-reads are L2 / Infinity-Cache hits and HBM sees every byte of x about once.
// ------------------------------------------------------------------------------------------------
template <int D, int SCAP, bool SEXACT, int WCAP, bool RREG>
__global__ void __launch_bounds__(256)
k_shifting(const float* __restrict__ x, long T, long C, const int4* __restrict__ year_plan, int n_cal, int W,
           int S_rt, int write_clim, const float* __restrict__ edges, int nb, long T_out, float* __restrict__ out,
           unsigned short* __restrict__ bins, unsigned char* __restrict__ mask, int* __restrict__ invalid_count,
           int ncb, int nchunks, int ablate, const int* __restrict__ skip) {
    extern __shared__ float lds[];
    // W-year history of every (cell, dayofyear).  LDS ring [D][WCAP][256] (slots W..WCAP-1 hold +0.0, neutral in
    // the sum), or -- RREG -- a register shift line per dayofyear: rr[i][WCAP-W .. WCAP-1] = years y-W .. y-1,
    // the leading WCAP-W entries stay +0.0.  The register line frees the LDS, so occupancy is set by VGPRs only.
    float* ring = lds;
    float rr[RREG ? D : 1][RREG ? WCAP : 1];
    const int npad = WCAP - W;
    int4* lplan = reinterpret_cast<int4*>(lds + (RREG ? (size_t)0 : (size_t)D * WCAP * 256));  // [n_cal][D] plan column
    float* e = reinterpret_cast<float*>(lplan + (size_t)n_cal * D);      // [nb+1] when binning

    int cb, chunk;
    if (!xcd_swizzle(blockIdx.x, ncb, nchunks, cb, chunk)) return;
    if (D == 4 && skip && skip[chunk]) return;  // this chunk of 4 dayofyears belongs to k_shift_fast
    const int tid = threadIdx.x;
    const long c = (long)cb * 256 + tid;
    const bool active = c < C;
    const int S = SEXACT ? SCAP : S_rt;
    const int lo = S / 2;
    const float Sf = (float)S;
    const int d0 = chunk * D;
    const bool do_bins = bins != nullptr;

    if (RREG) {
#pragma unroll
        for (int i = 0; i < D; ++i)
#pragma unroll
            for (int j = 0; j < WCAP; ++j) rr[i][j] = j >= npad ? nan_f() : 0.f;
    } else if (!(ablate & 128)) {
        for (int i = tid; i < D * WCAP * 256; i += 256) ring[i] = ((i >> 8) % WCAP) < W ? nan_f() : 0.f;
    }
    // ring_read: the W history values in ascending year order plus the +0.0 pads (LDS: pads last, RREG: pads first;
    // adding +0.0 before or after changes nothing: the sum starts at +0.0).  ring_push: year y replaces year y-W.
    auto ring_read = [&](int i, int slot0, float (&rv)[WCAP]) {
        if (RREG) {
#pragma unroll
            for (int j = 0; j < WCAP; ++j) rv[j] = rr[i][j];
        } else {
            const float* col = ring + (size_t)i * WCAP * 256 + tid;
#pragma unroll
            for (int j = 0; j < WCAP; ++j) {
                int sl = slot0 + j;
                sl = sl >= W ? sl - W : sl;
                sl = j < W ? sl : j;
                rv[j] = col[sl * 256];
            }
        }
    };
    auto ring_push = [&](int i, int slot0, float v) {
        if (RREG) {
#pragma unroll
            for (int j = 0; j < WCAP - 1; ++j) rr[i][j] = j >= npad ? rr[i][j + 1] : 0.f;
            rr[i][WCAP - 1] = v;
        } else {
            ring[((size_t)i * WCAP + slot0) * 256 + tid] = v;
        }
    };
    auto is_real = [&](int j) { return RREG ? j >= npad : j < W; };
    // the chunk's {timestep, output row, bin row} entries of every year, staged once: the year loop then reads
    // them from LDS instead of waiting on a scalar global load per year (dayofyears past 366 count as absent)
    for (int i = tid; i < n_cal * D; i += 256) {
        const int y = i / D, k = i - y * D;
        lplan[i] = (d0 + k < NDOY) ? year_plan[(size_t)y * NDOY + d0 + k] : make_int4(-1, -1, -1, 0);
    }
    if (do_bins)
        for (int i = tid; i <= nb; i += 256) e[i] = edges[i];
    __syncthreads();
    float inv_width = 0.f, e_first = 0.f, e_delta = 0.f, e_last = 0.f;
    bool arange_tab = false;
    if (do_bins) {
        inv_width = (float)(nb - 1) / (e[nb] - e[1]);
        e_first = e[1];
        e_delta = e[2] - e[1];
        e_last = e[nb];
        arange_tab = edges_are_arange(e, nb) && !(ablate & 512);
    }

    int n_invalid = 0;
    if (chunk == 0 && mask && active) mask[c] = finite_f(x[c]) ? 1 : 0;
    if (ablate & 256) return;  // timing only: launch + init
    // Inactive lanes (beyond C) stream -- and store -- the last cell instead of being masked off: loads and
    // stores stay unconditional and uniform in control flow (duplicate stores write identical values).
    const unsigned cidx = active ? (unsigned)c : (unsigned)(C - 1);

    // One (dayofyear i, year y) element: climatology from the ring, anomaly, bin, then push s into the ring.
    // pl = {timestep, output row, bin-matrix row} (-1: none).  The ring is read with NO predicates: the slots
    // of years y-W .. y-1 in ascending order (slot0, slot0+1, ... mod W), then the +0.0 pad slots.
    auto emit = [&](int i, int slot0, const int4 pl, float xc, float s) {
        if (ablate & 1) {  // timing only: keep the inputs alive, skip climatology / anomaly / bins / stores
            if (xc + s == 12345.678f) out[cidx] = xc;
            return;
        }
        if (pl.x >= 0) {  // uniform
            n_invalid += finite_f(xc) ? 0 : 1;
            if (pl.y >= 0) {  // uniform: this timestep is an output row
                float rv[WCAP];
                float acc = 0.f;
                int n = W;
                if (ablate & 8) {
                    acc = s;
                } else {
                    ring_read(i, slot0, rv);
#pragma unroll
                    for (int j = 0; j < WCAP; ++j) acc += rv[j];
                    // a NaN term (leap day, first days of the series, gaps): redo as nanmean.  Land lanes (NaN
                    // centre value) never need it -- their anomaly is NaN whatever the climatology is.
                    if (!(acc == acc) && (write_clim || xc == xc)) {
                        acc = 0.f;
                        n = 0;
#pragma unroll
                        for (int j = 0; j < WCAP; ++j) {
                            if (is_real(j) && rv[j] == rv[j]) {
                                acc += rv[j];
                                ++n;
                            }
                        }
                    }
                }
                const float clim = (ablate & 16) ? acc : acc / (float)n;  // n == 0 -> 0/0 = NaN
                const float a = xc - clim;
                // lanes beyond C duplicate the last cell (same inputs, same values): stores need no guard
                if (!(ablate & 64)) out[(size_t)pl.y * C + cidx] = write_clim ? clim : a;
                if (do_bins && !(ablate & 32))
                    bins[(ablate & 1024) ? (size_t)cidx : bins_index(pl.z, cidx, T_out)] = (unsigned short)(
                        arange_tab ? digitize_arange(a, e_first, e_delta, e_last, nb, inv_width)
                                   : digitize_bin(a, e, nb, inv_width));
                if ((ablate & 96) == 96 && a == 12345.678f) out[cidx] = a;
            }
        }
        ring_push(i, slot0, (pl.x >= 0) ? s : nan_f());
    };

    // The same for all D dayofyears of a year at once, phase by phase, so that the D independent dependency
    // chains (ring sums, divisions, bin search) overlap instead of running one after the other.  Used when every
    // one of the D timesteps is an output row (the common case after the first W years).
    auto emit_all = [&](int slot0, const int4 (&pl)[D], const float (&xc)[D], const float (&sm)[D]) {
        float rv[D][WCAP];
#pragma unroll
        for (int i = 0; i < D; ++i) ring_read(i, slot0, rv[i]);
        float acc[D];
        int n[D];
        bool slow = false;
#pragma unroll
        for (int i = 0; i < D; ++i) {
            acc[i] = 0.f;
            n[i] = W;
            n_invalid += finite_f(xc[i]) ? 0 : 1;
        }
#pragma unroll
        for (int j = 0; j < WCAP; ++j)
#pragma unroll
            for (int i = 0; i < D; ++i) acc[i] += rv[i][j];
#pragma unroll
        for (int i = 0; i < D; ++i) slow |= !(acc[i] == acc[i]) && (write_clim || xc[i] == xc[i]);
        if (slow) {  // a NaN term somewhere (leap day, first days of the series, gaps): nanmean for those
#pragma unroll
            for (int i = 0; i < D; ++i) {
                if (!(acc[i] == acc[i]) && (write_clim || xc[i] == xc[i])) {
                    acc[i] = 0.f;
                    n[i] = 0;
#pragma unroll
                    for (int j = 0; j < WCAP; ++j) {
                        if (is_real(j) && rv[i][j] == rv[i][j]) {
                            acc[i] += rv[i][j];
                            ++n[i];
                        }
                    }
                }
            }
        }
        float a[D], clim[D];
#pragma unroll
        for (int i = 0; i < D; ++i) {
            clim[i] = acc[i] / (float)n[i];  // n == 0 -> 0/0 = NaN
            a[i] = xc[i] - clim[i];
        }
#pragma unroll
        for (int i = 0; i < D; ++i) out[(size_t)pl[i].y * C + cidx] = write_clim ? clim[i] : a[i];
        if (do_bins) {
            int kb[D];
#pragma unroll
            for (int i = 0; i < D; ++i)
                kb[i] = arange_tab ? digitize_arange(a[i], e_first, e_delta, e_last, nb, inv_width)
                                   : digitize_bin(a[i], e, nb, inv_width);
#pragma unroll
            for (int i = 0; i < D; ++i) bins[bins_index(pl[i].z, cidx, T_out)] = (unsigned short)kb[i];
        }
#pragma unroll
        for (int i = 0; i < D; ++i) ring_push(i, slot0, sm[i]);
    };

    constexpr int NL = D + SCAP - 1;
    float xw[NL];
    for (int y = 0; y < n_cal; ++y) {
        // this year's D plan entries (dayofyears past 366 in the last chunk count as absent)
        int4 pl[D];
        bool fast = true, any = false;
#pragma unroll
        for (int i = 0; i < D; ++i) {
            pl[i] = lplan[y * D + i];
            any |= pl[i].x >= 0;
            fast &= (pl[i].x >= 0) && (pl[i].x == pl[0].x + i);
        }
        if (!RREG && !any && y < W) continue;  // nothing to read, and the ring slot of this year still holds its initial NaN
        const int slot0 = y % W;
        const long r0 = (long)pl[0].x - lo;
        if (fast && SEXACT && r0 >= 0 && r0 + NL <= T) {
            // D consecutive timesteps, whole batch inside the series: NL unconditional row loads, no predicates
            if (ablate & 4) {
#pragma unroll
                for (int j = 0; j < NL; ++j) xw[j] = (float)(j + y) * 0.25f;
            } else {
                const float* rowp = x + (size_t)r0 * C;
#pragma unroll
                for (int j = 0; j < NL; ++j) {
                    xw[j] = rowp[cidx];
                    rowp += C;
                }
            }
            float sacc[D], xcen[D], smo[D];
            bool all_out = !ablate;
#pragma unroll
            for (int i = 0; i < D; ++i) {
                sacc[i] = xw[i];
                xcen[i] = xw[i + SCAP / 2];
                all_out &= pl[i].y >= 0;
            }
            if (ablate & 2) {
#pragma unroll
                for (int i = 0; i < D; ++i) sacc[i] = xw[i] + xw[NL - 1];
            } else {
#pragma unroll
                for (int k = 1; k < SCAP; ++k)
#pragma unroll
                    for (int i = 0; i < D; ++i) sacc[i] += xw[i + k];
            }
#pragma unroll
            for (int i = 0; i < D; ++i) smo[i] = sacc[i] / Sf;
            if (all_out) {
                emit_all(slot0, pl, xcen, smo);
            } else {
#pragma unroll
                for (int i = 0; i < D; ++i) emit(i, slot0, pl[i], xcen[i], smo[i]);
            }
        } else {
            // generic path: every present dayofyear loads its own S rows with range checks (series ends,
            // calendar gaps, leap day, runtime S)
#pragma unroll
            for (int i = 0; i < D; ++i) {
                float xc = nan_f(), sm = nan_f();
                if (pl[i].x >= 0) {
                    const long q0 = (long)pl[i].x - lo;
                    float acc = 0.f;
                    for (int k = 0; k < S; ++k) {
                        const long row = q0 + k;
                        const float v = (row >= 0 && row < T) ? x[(size_t)row * C + cidx] : nan_f();
                        acc = (k == 0) ? v : acc + v;
                        if (k == lo) xc = v;
                    }
                    sm = acc / Sf;
                }
                emit(i, slot0, pl[i], xc, sm);
            }
        }
    }
    if (invalid_count && active && n_invalid) atomicAdd(&invalid_count[c], n_invalid);
}

// ------------------------------------------------------------------------------------------------
// K_A fast path: the same arithmetic as k_shifting for the regular part of the calendar, written for
// the VALU (which is what bounds k_shifting: 386 vector instructions per wave, year and 4 dayofyears).
//
//   * wave = 64 cells x 4 consecutive dayofyears; the 4 waves of a workgroup take 4 neighbouring
//     chunks of the SAME 64 cells.  Everything about the calendar is wave-uniform and lives in SGPRs
//     (the plan is read with scalar loads one year ahead), so row addresses are SGPR base + lane offset.
//   * two dayofyears per instruction: packed float32 (v_pk_add/mul/fma_f32).  The two smoothing chains
//     of a pair are skewed by one step so that both add the SAME row in one instruction (op_sel
//     broadcast) -- 23 instructions for the 40 sequential adds of a pair, bit-identical to the
//     sequential sums (the lagging chain starts from -0.0, the identity of IEEE addition).
//   * "/ S" and "/ W" are a * fl(1/b) followed by one Markstein correction step (two fma) and
//     v_div_fixup_f32 for zeros / infinities: bit-identical to IEEE division for every float32 a when b
//     is odd or a power of two (exhaustive check over b <= 64: oracle/proofs/div_by_const.c; even b
//     have ties among subnormal quotients and keep the real division).
//   * np.digitize on an arange table: one fused guess, two edges recomputed with the table's own
//     arithmetic, +-1 correction (classify() proves the guess is within one bin before enabling this).
//   * the W-year history is a register shift line of exactly W packed pairs (template parameter).
// Chunks the calendar makes irregular (leap day, gaps, series starting mid-chunk) and every other
// configuration (S != 21, W > 16, arbitrary edge tables) stay on k_shifting; k_shift_classify decides
// per chunk on the device, both kernels skip the other's chunks.
// ------------------------------------------------------------------------------------------------
typedef float v2f __attribute__((ext_vector_type(2)));

// acc.lo += src.lo, acc.hi += src.lo  /  acc.lo += src.hi, acc.hi += src.hi
__device__ __forceinline__ v2f pk_add_bc_lo(v2f acc, v2f src) {
    v2f r;
    asm("v_pk_add_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(r) : "v"(acc), "v"(src));
    return r;
}
__device__ __forceinline__ v2f pk_add_bc_hi(v2f acc, v2f src) {
    v2f r;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1]" : "=v"(r) : "v"(acc), "v"(src));
    return r;
}
__device__ __forceinline__ v2f splat2(float v) { return (v2f){v, v}; }

// a / b for a constant b (y = fl(1/b)): bit-identical to IEEE division for odd b and powers of two up to 64
// (exhaustive over all 2^32 a: oracle/proofs/div_by_const.c; v_div_fixup supplies the +-0 and +-inf cases)
__device__ __forceinline__ v2f div_const2(v2f a, float b, float y) {
    const v2f q = a * splat2(y);
    const v2f r = __builtin_elementwise_fma(-q, splat2(b), a);
    const v2f q2 = __builtin_elementwise_fma(r, splat2(y), q);
    return (v2f){__builtin_amdgcn_div_fixupf(q2.x, b, a.x), __builtin_amdgcn_div_fixupf(q2.y, b, a.y)};
}

// Buffer addressing: 128-bit descriptor in SGPRs (wave-uniform base), 32-bit lane byte offset in a VGPR, 32-bit
// uniform byte offset in an SGPR -- no vector instruction is spent on addresses.
typedef __amdgpu_buffer_rsrc_t rsrc_t;
__device__ __forceinline__ rsrc_t make_rsrc(const void* base) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, -1, 0x00020000);
}
__device__ __forceinline__ float ldb_f32(rsrc_t r, unsigned voff, int soff) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int)voff, soff, 0));
}
// cache policy of the anomaly stores: non-temporal (bit 1) -- the rows are never read again by this kernel, keeping
// them out of the L2 leaves it to the input rows that neighbouring workgroups re-read (measured -5 % on a 100-yr band;
// the same hint on the 2-byte bin stores is 15 % slower)
#ifndef ST_AUX_F32
#define ST_AUX_F32 2
#endif
#ifndef ST_AUX_U16
#define ST_AUX_U16 0
#endif
__device__ __forceinline__ void stb_f32(rsrc_t r, unsigned voff, int soff, float v) {
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), r, (int)voff, soff, ST_AUX_F32);
}
__device__ __forceinline__ void stb_u16(rsrc_t r, unsigned voff, int soff, int v) {
    __builtin_amdgcn_raw_buffer_store_b16((unsigned short)v, r, (int)voff, soff, ST_AUX_U16);
}

#define SHIFT_INFO_WORDS 128  // [0..91] chunk handled by the fast kernel, [92] arange edges usable

__global__ void k_shift_classify(const int4* __restrict__ year_plan, int n_cal, const float* __restrict__ edges, int nb,
                                 int want_bins, int enable, int* __restrict__ info) {
    __shared__ int s_edges_ok;
    const int t = threadIdx.x;
    if (t == 0) {
        int ok = 1;
        if (want_bins) {
            const float first = edges[1], delta = edges[2] - edges[1], last = edges[nb];
            ok = delta > 0.f && nb < 32768;
            for (int j = 1; ok && j <= nb; ++j)
                ok = __float_as_uint(edges[j]) == __float_as_uint(arange_edge(j, first, delta));
            // the fused guess (biased down by 1/128 bin) must land in the true bin or the one below: generous bound
            // on its rounding error (about 8x what the individual roundings add up to)
            const double m = fabs((double)first) > fabs((double)last) ? fabs((double)first) : fabs((double)last);
            ok = ok && ((double)nb + 2.0 * m / (double)delta) * (1.0 / 1048576.0) < 1.0 / 256.0;
        }
        s_edges_ok = ok && enable;
        info[92] = s_edges_ok;
    }
    __syncthreads();
    if (t >= 92) return;
    const int d0 = t * 4;
    int ok = s_edges_ok;
    // every year: the first m (0..4) dayofyears of the chunk present on consecutive timesteps, the rest absent
    // (leap day; dayofyears past 366 in the last chunk); output rows all or none, consecutive
    for (int y = 0; ok && y < n_cal; ++y) {
        int4 e[4];
        for (int i = 0; i < 4; ++i)
            e[i] = (d0 + i < NDOY) ? year_plan[(size_t)y * NDOY + d0 + i] : make_int4(-1, -1, -1, 0);
        int m = 0;
        while (m < 4 && e[m].x >= 0) ++m;
        for (int i = m; i < 4; ++i) ok = ok && e[i].x < 0;
        for (int i = 1; i < m; ++i) {
            ok = ok && e[i].x == e[0].x + i;
            if (e[0].y >= 0)
                ok = ok && e[i].y == e[0].y + i && e[i].z >= 0;
            else
                ok = ok && e[i].y < 0;
        }
        if (m > 0 && e[0].y >= 0) ok = ok && e[0].z >= 0;
    }
    info[t] = ok;
}

template <int W>
__global__ void __launch_bounds__(256)
k_shift_fast(const float* __restrict__ x, long T, long C, const int4* __restrict__ year_plan, int n_cal,
             const int* __restrict__ info, int write_clim, const float* __restrict__ edges, int nb, long T_out, float* __restrict__ out, unsigned short* __restrict__ bins, unsigned char* __restrict__ mask,
             int* __restrict__ invalid_count, int ncg, int nblk) {
    int cg, bc;
    if (!xcd_swizzle(blockIdx.x, ncg, nblk, cg, bc)) return;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int lane = (int)(threadIdx.x & 63);
    const int chunk = bc * 4 + wave;
    // The 4 waves work on 4 neighbouring chunks of the same 64 cells: the 36 rows one calendar year needs for
    // all 16 dayofyears are staged once in LDS (double buffered, loaded one year ahead) and every wave reads its
    // 24 from there -- 2.25 instead of 6 row reads per output row leave the L2.
    __shared__ float stage[2][36 * 64];
    const bool mine = chunk < 92 && info[chunk] != 0;  // wave-uniform; the other waves only help staging
    const int d0 = mine ? chunk * 4 : 0;
    const int4* pblk = year_plan + bc * 16;  // first dayofyear of the workgroup (bc <= 22: always < 366)
    const long c = (long)cg * 64 + lane;
    const bool active = c < C;
    const unsigned cidx = active ? (unsigned)c : (unsigned)(C - 1);  // lanes beyond C duplicate the last cell
    const bool do_bins = bins != nullptr;
    // bin matrix: lane part of the element index relative to the wave's first 16-cell block
    const unsigned voff = cidx * 4u;  // byte offset of the lane's cell inside a (time, cell) row
    const int rowb = (int)(C * 4);    // bytes per (time, cell) row
    const unsigned bin_lane = (((cidx >> 4) - (unsigned)(cg * 4)) * (unsigned)T_out * 16u + (cidx & 15u)) * 2u;  // bytes
    const rsrc_t rbins = make_rsrc(do_bins ? bins + (size_t)(cg * 4) * (size_t)T_out * 16 : nullptr);

    float e_first = 0.f, e_delta = 1.f, inv_width = 1.f;
    if (do_bins) {
        e_first = edges[1];
        e_delta = edges[2] - edges[1];
        inv_width = (float)(nb - 1) / (edges[nb] - e_first);
    }
    constexpr float Sf = 21.f;
    const float yS = 1.0f / Sf;
    const float Wf = (float)W;
    const float yW = 1.0f / Wf;
    const float nbm1f = (float)(nb - 1);
    const float c0 = (1.0f - e_first * inv_width) - 0.0078125f;  // +1 (edges[0] = -inf) and the 1/128-bin downward bias
    const float qnan = nan_f();

    if (chunk == 0 && mine && mask && active) mask[c] = finite_f(x[c]) ? 1 : 0;
    // rows tb-10 .. tb+25 (tb = timestep of the workgroup's first dayofyear in that year) can be staged when they
    // all lie inside the series; this wave loads rows 9*wave .. 9*wave+8 of them
    auto stage_ok = [&](int tb) { return tb >= 10 && (long)tb + 26 <= T; };
    auto stage_load = [&](int tb, float (&nx)[9]) {
        const rsrc_t rs = make_rsrc(x + (size_t)(tb - 10 + 9 * wave) * C);
#pragma unroll
        for (int k = 0; k < 9; ++k) nx[k] = ldb_f32(rs, voff, k * rowb);
    };
    auto stage_store = [&](int buf, const float (&nx)[9]) {
#pragma unroll
        for (int k = 0; k < 9; ++k) stage[buf][(9 * wave + k) * 64 + lane] = nx[k];
    };
    int tb_next = pblk[0].x;
    {
        float nx[9];
        if (stage_ok(tb_next)) {
            stage_load(tb_next, nx);
            stage_store(0, nx);
        }
    }
    __syncthreads();

    // History of dayofyears (0,1) and (2,3) as register lines.  The year loop is unrolled by two: the first year of a
    // pair reads entries [0, W) and appends at [W], the second reads [1, W] and appends at [W+1], then the line moves
    // down by two -- W moves per two years instead of 2 (W - 1).
    v2f rA[W + 2], rB[W + 2];
#pragma unroll
    for (int j = 0; j < W + 2; ++j) rA[j] = rB[j] = splat2(qnan);
    int n_invalid = 0;

    const int4* pp = year_plan + d0;
    const bool tail = d0 + 3 >= NDOY;  // last chunk: dayofyears 365, 366 and two that do not exist
    const int4 absent = make_int4(-1, -1, -1, 0);
    int4 n0 = pp[0], n1 = pp[1], n2 = tail ? absent : pp[2], n3 = tail ? absent : pp[3];
    auto one_year = [&](int y, auto Jc) {
        constexpr int J = decltype(Jc)::value;
        const int4 p0 = n0, p1 = n1, p2 = n2, p3 = n3;
        const int tb = tb_next;
        float nx[9];
        bool stage_next = false;
        if (y + 1 < n_cal) {  // next year's plan and rows, one iteration ahead
            const int4* q = pp + (size_t)(y + 1) * NDOY;
            n0 = q[0];
            n1 = q[1];
            n2 = tail ? absent : q[2];
            n3 = tail ? absent : q[3];
            tb_next = pblk[(size_t)(y + 1) * NDOY].x;
            stage_next = stage_ok(tb_next);
            if (stage_next) stage_load(tb_next, nx);
        }
        v2f smA = splat2(qnan), smB = splat2(qnan);
        if (mine && p0.x >= 0) {
            const long r0 = (long)p0.x - 10;
            v2f xp[12];  // xp[m] = rows (r0 + 2m, r0 + 2m + 1)
            const bool edge = r0 < 0 || r0 + 24 > T;
            if (stage_ok(tb) && p0.x == tb + 4 * wave) {
                const float* st = &stage[y & 1][(4 * wave) * 64 + lane];
#pragma unroll
                for (int m = 0; m < 12; ++m) {
                    xp[m].x = st[(2 * m) * 64];
                    xp[m].y = st[(2 * m + 1) * 64];
                }
            } else if (!edge) {
                const rsrc_t rx = make_rsrc(x + (size_t)r0 * C);
#pragma unroll
                for (int m = 0; m < 12; ++m) {
                    xp[m].x = ldb_f32(rx, voff, (2 * m) * rowb);
                    xp[m].y = ldb_f32(rx, voff, (2 * m + 1) * rowb);
                }
            } else {
#pragma unroll
                for (int m = 0; m < 12; ++m) {
                    long ra = r0 + 2 * m, rb = ra + 1;
                    ra = ra < 0 ? 0 : (ra > T - 1 ? T - 1 : ra);
                    rb = rb < 0 ? 0 : (rb > T - 1 ? T - 1 : rb);
                    xp[m].x = ldb_f32(make_rsrc(x + (size_t)ra * C), voff, 0);
                    xp[m].y = ldb_f32(make_rsrc(x + (size_t)rb * C), voff, 0);
                }
            }
            // smoothing: sequential sums of rows i .. i+20 for the four dayofyears i = 0..3
            v2f accA = (v2f){xp[0].x, -0.0f};
#pragma unroll
            for (int s = 1; s <= 20; ++s) accA = (s & 1) ? pk_add_bc_hi(accA, xp[s >> 1]) : pk_add_bc_lo(accA, xp[s >> 1]);
            accA.y += xp[10].y;
            v2f accB = (v2f){xp[1].x, -0.0f};
#pragma unroll
            for (int s = 3; s <= 22; ++s) accB = (s & 1) ? pk_add_bc_hi(accB, xp[s >> 1]) : pk_add_bc_lo(accB, xp[s >> 1]);
            accB.y += xp[11].y;
            smA = div_const2(accA, Sf, yS);
            smB = div_const2(accB, Sf, yS);
            if (edge) {  // windows that leave the series: NaN (a NaN row in the sum, in the general kernel)
                const long t0 = p0.x;
                smA.x = (t0 - 10 >= 0 && t0 + 10 < T) ? smA.x : qnan;
                smA.y = (t0 - 9 >= 0 && t0 + 11 < T) ? smA.y : qnan;
                smB.x = (t0 - 8 >= 0 && t0 + 12 < T) ? smB.x : qnan;
                smB.y = (t0 - 7 >= 0 && t0 + 13 < T) ? smB.y : qnan;
            }
            const v2f xcA = xp[5], xcB = xp[6];
            const bool partial = p3.x < 0;  // only a prefix of the 4 dayofyears exists this year (leap day chunk)
            if (!partial) {
                n_invalid += (finite_f(xcA.x) ? 0 : 1) + (finite_f(xcA.y) ? 0 : 1) + (finite_f(xcB.x) ? 0 : 1) +
                             (finite_f(xcB.y) ? 0 : 1);
            } else {
                n_invalid += finite_f(xcA.x) ? 0 : 1;
                if (p1.x >= 0) n_invalid += finite_f(xcA.y) ? 0 : 1; else smA.y = qnan;
                if (p2.x >= 0) n_invalid += finite_f(xcB.x) ? 0 : 1; else smB.x = qnan;
                smB.y = qnan;
            }
            if (p0.y >= 0) {  // output rows
                v2f sA = splat2(0.f), sB = splat2(0.f);
#pragma unroll
                for (int j = 0; j < W; ++j) {
                    sA = sA + rA[J + j];
                    sB = sB + rB[J + j];
                }
                // the reciprocal form is exact for odd W and powers of two only (even W have halfway cases among
                // subnormal quotients that it misrounds: oracle/proofs/div_by_const.c); other W divide for real
                constexpr bool recip_exact = (W & 1) || (W & (W - 1)) == 0;
                v2f climA, climB;
                if (recip_exact) {
                    climA = div_const2(sA, Wf, yW);
                    climB = div_const2(sB, Wf, yW);
                } else {
                    climA = (v2f){sA.x / Wf, sA.y / Wf};
                    climB = (v2f){sB.x / Wf, sB.y / Wf};
                }
                // a NaN in the history (first days of the series, gaps) while the centre value is a number: nanmean
                const bool slow = (!(climA.x == climA.x) && (write_clim || xcA.x == xcA.x)) ||
                                  (!(climA.y == climA.y) && (write_clim || xcA.y == xcA.y)) ||
                                  (!(climB.x == climB.x) && (write_clim || xcB.x == xcB.x)) ||
                                  (!(climB.y == climB.y) && (write_clim || xcB.y == xcB.y));
                if (__builtin_amdgcn_ballot_w64(slow) != 0) {
                    float acc[4] = {0.f, 0.f, 0.f, 0.f};
                    int n[4] = {0, 0, 0, 0};
#pragma unroll
                    for (int j = 0; j < W; ++j) {
                        const float v[4] = {rA[J + j].x, rA[J + j].y, rB[J + j].x, rB[J + j].y};
#pragma unroll
                        for (int i = 0; i < 4; ++i)
                            if (v[i] == v[i]) {
                                acc[i] += v[i];
                                ++n[i];
                            }
                    }
                    // elements without a NaN term keep the fast result (identical: same sum, n == W)
                    if (!(climA.x == climA.x)) climA.x = acc[0] / (float)n[0];
                    if (!(climA.y == climA.y)) climA.y = acc[1] / (float)n[1];
                    if (!(climB.x == climB.x)) climB.x = acc[2] / (float)n[2];
                    if (!(climB.y == climB.y)) climB.y = acc[3] / (float)n[3];
                }
                const v2f aA = xcA - climA, aB = xcB - climB;
                const rsrc_t ro = make_rsrc(out + (size_t)p0.y * C);
                stb_f32(ro, voff, 0, write_clim ? climA.x : aA.x);
                if (p1.x >= 0) stb_f32(ro, voff, rowb, write_clim ? climA.y : aA.y);
                if (p2.x >= 0) stb_f32(ro, voff, 2 * rowb, write_clim ? climB.x : aB.x);
                if (p3.x >= 0) stb_f32(ro, voff, 3 * rowb, write_clim ? climB.y : aB.y);
                if (do_bins) {
                    // np.digitize(a, edges) - 1 on the arange table (contract C4): the guess, biased down, is the
                    // true bin or the one below (k_shift_classify checked the error bound); one comparison with
                    // the edge above it -- recomputed with the table's own arithmetic -- settles which.  A NaN
                    // guess clamps to 0 and is replaced by nb at the end.
                    auto digit2 = [&](v2f a, int& k0, int& k1) {
                        const v2f f = __builtin_elementwise_fma(a, splat2(inv_width), splat2(c0));
                        v2f t;
                        t.x = __builtin_amdgcn_fmed3f(__builtin_floorf(f.x), 0.0f, nbm1f);
                        t.y = __builtin_amdgcn_fmed3f(__builtin_floorf(f.y), 0.0f, nbm1f);
                        const v2f phi = t * splat2(e_delta);
                        const v2f ehi = splat2(e_first) + phi;  // edges[t + 1]
                        k0 = (int)t.x + (a.x >= ehi.x ? 1 : 0);
                        k1 = (int)t.y + (a.y >= ehi.y ? 1 : 0);
                        k0 = (a.x == a.x) ? k0 : nb;
                        k1 = (a.y == a.y) ? k1 : nb;
                    };
                    int k0, k1, k2, k3;
                    digit2(aA, k0, k1);
                    digit2(aB, k2, k3);
                    stb_u16(rbins, bin_lane, p0.z * 32, k0);
                    if (p1.x >= 0) stb_u16(rbins, bin_lane, p1.z * 32, k1);
                    if (p2.x >= 0) stb_u16(rbins, bin_lane, p2.z * 32, k2);
                    if (p3.x >= 0) stb_u16(rbins, bin_lane, p3.z * 32, k3);
                }
            }
        }
        rA[J + W] = smA;  // year y joins the history
        rB[J + W] = smB;
        if (stage_next) stage_store((y + 1) & 1, nx);
        __syncthreads();
    };
    for (int y = 0; y < n_cal; y += 2) {
        one_year(y, std::integral_constant<int, 0>{});
        if (y + 1 < n_cal) one_year(y + 1, std::integral_constant<int, 1>{});
#pragma unroll
        for (int j = 0; j < W; ++j) {
            rA[j] = rA[j + 2];
            rB[j] = rB[j + 2];
        }
    }
    if (invalid_count && active && n_invalid) atomicAdd(&invalid_count[c], n_invalid);
}

struct ShiftArgs {
    const float* x;
    int64_t T, C;
    const int4* year_plan;
    int n_cal;
    int W, S, write_clim;
    const float* edges;
    int nb;
    int64_t T_out;
    float* out;
    uint16_t* bins;
    uint8_t* mask;
    int32_t* invalid_count;
    const int* skip;
};

template <int D, int SCAP, bool SEXACT, int WCAP, bool RREG>
static int launch_shifting(marex_ctx* ctx, const ShiftArgs& a) {
    const int ncb = (int)((a.C + 255) / 256);
    const int nchunks = (NDOY + D - 1) / D;
    const size_t lds = ((RREG ? (size_t)0 : (size_t)D * WCAP * 256) + (a.bins ? (size_t)a.nb + 1 : 0)) * sizeof(float) + (size_t)a.n_cal * D * 16;
    if (lds > 80 * 1024) return fail(ctx, -4, "marex_shifting_baseline_f32: window_year_baseline=%d needs more than 80 KiB of LDS", a.W);
    auto kern = k_shifting<D, SCAP, SEXACT, WCAP, RREG>;
    if (lds > 48 * 1024)
        HIP_TRY(ctx, hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(kern, dim3(xcd_grid(ncb, nchunks)), dim3(256), lds, ctx->stream, a.x, (long)a.T, (long)a.C,
                       a.year_plan, a.n_cal, a.W, a.S, a.write_clim, a.edges, a.nb, (long)a.T_out, a.out, a.bins,
                       a.mask, a.invalid_count, ncb, nchunks, env_int("MAREX_SHIFT_ABLATE", 0), D == 4 ? a.skip : nullptr);
    HIP_TRY(ctx, hipGetLastError());
    return 0;
}

template <int W>
static void launch_shift_fast(marex_ctx* ctx, const ShiftArgs& a) {
    const int ncg = (int)((a.C + 63) / 64);
    hipLaunchKernelGGL(k_shift_fast<W>, dim3(xcd_grid(ncg, 23)), dim3(256), 0, ctx->stream, a.x, (long)a.T, (long)a.C,
                       a.year_plan, a.n_cal, a.skip, a.write_clim, a.edges, a.nb, (long)a.T_out, a.out, a.bins, a.mask,
                       a.invalid_count, ncg, 23);
}

template <int D, int WCAP, bool RREG>
static int dispatch_shifting_S(marex_ctx* ctx, const ShiftArgs& a) {
    if (a.S == 21) return launch_shifting<D, 21, true, WCAP, RREG>(ctx, a);
    return launch_shifting<D, 1, false, WCAP, RREG>(ctx, a);  // any other smoothing width: generic row loop
}

extern "C" int marex_shifting_baseline_f32(marex_ctx* ctx, const float* x, int64_t T, int64_t C,
                                           const int32_t* year_plan, int n_cal_years, int W, int S,
                                           int write_clim, const float* edges, int nb, int64_t T_out, float* out,
                                           uint16_t* bins, uint8_t* mask, int32_t* invalid_count) {
    if (!ctx) return -1;
    if (!x || !year_plan || !out || T <= 0 || C <= 0 || n_cal_years <= 0)
        return fail(ctx, -1, "marex_shifting_baseline_f32: null pointer or empty shape");
    if (((uintptr_t)year_plan & 15) != 0) return fail(ctx, -1, "marex_shifting_baseline_f32: year_plan must be 16-byte aligned");
    if (W < 1 || S < 1) return fail(ctx, -1, "marex_shifting_baseline_f32: W and S must be >= 1");
    if (S > T) S = (int)T + 1;  // every window leaves the series: all-NaN smoothing either way
    if (W > 64) return fail(ctx, -4, "marex_shifting_baseline_f32: window_year_baseline > 64 is not supported");
    if (bins && (!edges || nb < 4 || nb > 65534 || T_out <= 0))
        return fail(ctx, -1, "marex_shifting_baseline_f32: binning needs edges, T_out and 4 <= nb <= 65534");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    ShiftArgs a{x, T, C, reinterpret_cast<const int4*>(year_plan), n_cal_years, W, S, write_clim,
                edges, nb, T_out, out, bins, mask, invalid_count, nullptr};
    // 4 dayofyears per workgroup (6 row loads per output) while the padded W-year LDS ring leaves room for two
    // workgroups per CU, otherwise one dayofyear
    const int forceD = env_int("MAREX_SHIFT_D", 0);
    const int reg = env_int("MAREX_SHIFT_RING", 1);  // 1 (default): history in registers, 0: LDS ring
    // regular chunks of the calendar go to k_shift_fast (S = 21, instantiated W, arange edge table)
    const bool fast_w = W == 3 || W == 4 || W == 5 || W == 6 || W == 7 || W == 10 || W == 13 || W == 15;
    const bool fast_cfg = env_int("MAREX_SHIFT_FAST", 1) != 0 && S == 21 && fast_w && forceD == 0 && T >= 24 &&
                          T_out < (1 << 24) && C < (1 << 24) && env_int("MAREX_SHIFT_ABLATE", 0) == 0;
    if (fast_cfg && !ctx->shift_info) HIP_TRY(ctx, hipMalloc((void**)&ctx->shift_info, SHIFT_INFO_WORDS * sizeof(int)));
    LaunchTimer lt(ctx, MAREX_K_SHIFTING);  // one timed region: classify + fast kernel + general kernel
    if (fast_cfg) {
        hipLaunchKernelGGL(k_shift_classify, dim3(1), dim3(128), 0, ctx->stream, a.year_plan, n_cal_years, edges, nb,
                           bins ? 1 : 0, 1, ctx->shift_info);
        a.skip = ctx->shift_info;
        switch (W) {
            case 3: launch_shift_fast<3>(ctx, a); break;
            case 4: launch_shift_fast<4>(ctx, a); break;
            case 5: launch_shift_fast<5>(ctx, a); break;
            case 6: launch_shift_fast<6>(ctx, a); break;
            case 7: launch_shift_fast<7>(ctx, a); break;
            case 10: launch_shift_fast<10>(ctx, a); break;
            case 13: launch_shift_fast<13>(ctx, a); break;
            default: launch_shift_fast<15>(ctx, a); break;
        }
        HIP_TRY(ctx, hipGetLastError());
    }
    if (W <= 8) {
        if (reg) return forceD == 2 ? dispatch_shifting_S<2, 8, true>(ctx, a) : forceD == 8 ? dispatch_shifting_S<8, 8, true>(ctx, a) : dispatch_shifting_S<4, 8, true>(ctx, a);
        return forceD == 1 ? dispatch_shifting_S<1, 8, false>(ctx, a) : dispatch_shifting_S<4, 8, false>(ctx, a);
    }
    if (W <= 16) {
        if (reg) return forceD == 2 ? dispatch_shifting_S<2, 16, true>(ctx, a) : dispatch_shifting_S<4, 16, true>(ctx, a);
        return forceD == 1 ? dispatch_shifting_S<1, 16, false>(ctx, a) : dispatch_shifting_S<4, 16, false>(ctx, a);
    }
    return dispatch_shifting_S<1, 64, false>(ctx, a);
}

// ------------------------------------------------------------------------------------------------
// K_T: day-of-year thresholds from pooled histograms
//
// One WAVE owns NW consecutive cells of one grid row and runs on its own (no workgroup barrier): it
// keeps the pooled (ws x ws cells, wd days) histogram of each of its cells in LDS and slides it over
// the day-of-year axis -- entering a day adds one dayofyear bucket of every cell of the neighbourhood
// and removes the one that leaves the window (integer counts, order independent => exact).  The
// quantile bin `iu` and the number of samples at or above it (`ge`) are tracked incrementally, so no
// pass over the nb bins is needed per day.  Counters are uint16 packed two per dword whenever the
// largest possible pooled count fits (PACK), halving LDS per cell and doubling the resident waves.
// ------------------------------------------------------------------------------------------------
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

template <bool PACK>
__device__ __forceinline__ unsigned hget(const unsigned* h, int b) {
    return PACK ? ((h[b >> 1] >> ((b & 1) * 16)) & 0xFFFFu) : h[b];
}
template <bool PACK>
__device__ __forceinline__ void hadd(unsigned* h, int b, int delta) {
    if (PACK)
        atomicAdd(&h[b >> 1], (unsigned)delta * (1u << ((b & 1) * 16)));
    else
        atomicAdd(&h[b], (unsigned)delta);
}

template <bool PACK>
__global__ void __launch_bounds__(256)
k_thresholds(const unsigned short* __restrict__ bins, long T_out, long C, int ny, int nx, int nseg_per_row, long nsegs,
             int NW, const int* __restrict__ doy_start, const float* __restrict__ first_anom,
             const float* __restrict__ centres, int nb, double q, int wd, int p, float lower_bound,
             float upper_bound, int row0, float* __restrict__ thr, marex_thr_stats* __restrict__ stats) {
    extern __shared__ unsigned lds_u[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const long seg = (long)blockIdx.x * 4 + wave;
    if (seg >= nsegs) return;  // no workgroup barrier anywhere below
    const int nbw = PACK ? (nb + 1) / 2 : nb;
    const int wave_words = NW * nbw + 3 * NW;
    unsigned* hist = lds_u + (size_t)wave * wave_words;  // [NW][nbw]
    int* st_iu = (int*)(hist + (size_t)NW * nbw);           // [NW] quantile bin
    int* st_ge = st_iu + NW;                                // [NW] samples with bin >= iu
    int* st_tot = st_ge + NW;                               // [NW] samples in the window

    const int j = (ny > 0) ? row0 + (int)(seg / nseg_per_row) : 0;
    const int i0 = (int)(seg % nseg_per_row) * NW;
    const int nout = (nx - i0) < NW ? (nx - i0) : NW;
    const int jlo = (j - p) < 0 ? 0 : j - p;
    const int jhi = (ny > 0) ? ((j + p) > ny - 1 ? ny - 1 : j + p) : 0;
    const int win = nout + 2 * p;  // input columns i0-p .. i0+nout-1+p (lon periodic)
    const int ncell_in = (jhi - jlo + 1) * win;
    const int pd = wd / 2;
    const int nslot = ncell_in >= 64 ? 1 : 64 / ncell_in;

    for (int i = lane; i < wave_words; i += 64) hist[i] = 0u;
    wave_sync();

    // Per-lane view of the neighbourhood, fixed for the whole day loop: which input cell(s) this lane
    // streams, where its bin column starts, and which of the wave's output cells it feeds.
    struct CellMap {
        long coloff;  // cell index into a bins row, -1: lane idle in this pass
        int o_lo, o_hi;
    };
    auto make_map = [&](int ic) {
        CellMap m;
        m.coloff = -1;
        m.o_lo = 0;
        m.o_hi = -1;
        if (ic >= 0 && ic < ncell_in) {
            const int rr = ic / win, ii = ic - rr * win;
            int gi = (i0 - p + ii) % nx;
            if (gi < 0) gi += nx;
            m.coloff = (long)(jlo + rr) * nx + gi;
            m.o_lo = (ii - 2 * p) < 0 ? 0 : ii - 2 * p;
            m.o_hi = ii < nout - 1 ? ii : nout - 1;
        }
        return m;
    };
    const int slot = nslot > 1 ? lane / ncell_in : 0;
    const int npass = nslot > 1 ? 1 : (ncell_in + 63) / 64;
    const CellMap map0 = make_map(nslot > 1 ? (slot < nslot ? lane % ncell_in : -1) : lane);
    const CellMap map1 = make_map(npass > 1 ? 64 + lane : -1);

    // One sample of a bucket: bins 0 / 1 (about half of all samples) are only counted here and
    // flushed once per lane and step; every other bin goes to the histograms of the fed cells.
    auto one_sample = [&](const CellMap& m, int b, int sgn, int& nvalid, int& n0, int& n1) {
        if (b >= nb) return;
        nvalid += sgn;
        if (b == 0) {
            n0 += sgn;
        } else if (b == 1) {
            n1 += sgn;
        } else {
            unsigned* h = hist + (size_t)m.o_lo * nbw;
            for (int o = m.o_lo; o <= m.o_hi; ++o, h += nbw) {
                hadd<PACK>(h, b, sgn);
                if (b >= st_iu[o]) atomicAdd(&st_ge[o], sgn);
            }
        }
    };
    // stream one bucket (rows r0 .. r0+nd-1 of the lane's column), 4 independent loads in flight
    auto stream_bucket = [&](const CellMap& m, int r0, int nd, int sgn, int& nvalid, int& n0, int& n1) {
        const unsigned short* col = bins + bins_index(r0, m.coloff, T_out);
        for (int r = slot; r < nd; r += 4 * nslot) {
            int bb[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int ru = r + u * nslot;
                bb[u] = ru < nd ? (int)col[(size_t)ru * 16] : nb;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) one_sample(m, bb[u], sgn, nvalid, n0, n1);
        }
    };
    auto flush = [&](const CellMap& m, int nvalid, int n0, int n1) {
        if ((nvalid | n0 | n1) == 0) return;
        unsigned* h = hist + (size_t)m.o_lo * nbw;
        for (int o = m.o_lo; o <= m.o_hi; ++o, h += nbw) {
            if (nvalid) atomicAdd(&st_tot[o], nvalid);
            if (PACK) {
                if (n0 | n1) atomicAdd(&h[0], (unsigned)(n0 + n1 * 65536));
            } else {
                if (n0) atomicAdd(&h[0], (unsigned)n0);
                if (n1) atomicAdd(&h[1], (unsigned)n1);
            }
            const int iu = st_iu[o];
            const int g = (iu <= 0 ? n0 : 0) + (iu <= 1 ? n1 : 0);
            if (g) atomicAdd(&st_ge[o], g);
        }
    };
    // enter dayofyear d_in (1-based, 0 = none) and leave d_out (0 = none) for the whole neighbourhood
    auto step_window = [&](int d_in, int d_out) {
        const int ri = d_in ? doy_start[d_in - 1] : 0, ni = d_in ? doy_start[d_in] - ri : 0;
        const int ro = d_out ? doy_start[d_out - 1] : 0, no = d_out ? doy_start[d_out] - ro : 0;
        for (int pass = 0; pass < npass; ++pass) {
            const CellMap m = pass == 0 ? map0 : (pass == 1 ? map1 : make_map(pass * 64 + lane));
            if (m.coloff < 0) continue;
            int nvalid = 0, n0 = 0, n1 = 0;
            stream_bucket(m, ri, ni, +1, nvalid, n0, n1);
            stream_bucket(m, ro, no, -1, nvalid, n0, n1);
            flush(m, nvalid, n0, n1);
        }
    };

    for (int o = -pd; o <= pd; ++o) step_window((o % NDOY + NDOY) % NDOY + 1, 0);
    wave_sync();

    unsigned kmin = 0xFFFFFFFFu, kmax = 0u, nlow = 0u, nhigh = 0u;
    const long cell = (long)j * nx + i0 + lane;
    bool land = true;
    if (lane < nout) land = !(first_anom[cell] == first_anom[cell]);
    for (int d = 0; d < NDOY; ++d) {
        if (d > 0) {
            step_window((d + pd) % NDOY + 1, ((d - pd - 1) % NDOY + NDOY) % NDOY + 1);
            wave_sync();
        }
        if (lane < nout) {
            const unsigned* h = hist + (size_t)lane * nbw;
            const int tot = st_tot[lane];
            int iu = st_iu[lane], ge = st_ge[lane];
            float t32 = nan_f();
            if (tot > 0) {
                const double qpos = q * (double)tot;
                while (iu < nb - 1) {
                    const int hv = (int)hget<PACK>(h, iu);
                    if (!((double)(tot - ge + hv) <= qpos)) break;
                    ge -= hv;
                    ++iu;
                }
                while (iu > 0 && (double)(tot - ge) > qpos) {
                    --iu;
                    ge += (int)hget<PACK>(h, iu);
                }
                if (!land) {
                    const int below = tot - ge;
                    const int il = iu > 0 ? iu - 1 : 0;
                    const int cs_iu = below + (int)hget<PACK>(h, iu);
                    const int cs_il = iu > 0 ? below : cs_iu;
                    const int diff = cs_iu - cs_il;
                    const double frac = diff > 0 ? (qpos - (double)cs_il) / (double)diff : 0.5;
                    const float dc = centres[iu] - centres[il];
                    const double prod = frac * (double)dc;
                    t32 = (float)((double)centres[il] + prod);
                    if (iu == 0) t32 = centres[0];
                }
            } else {
                iu = 0;
                ge = 0;
            }
            st_iu[lane] = iu;
            st_ge[lane] = ge;
            if (t32 == t32) {
                const unsigned k = ordered_key(t32);
                kmin = k < kmin ? k : kmin;
                kmax = k > kmax ? k : kmax;
                if (t32 > upper_bound) ++nhigh;
                if (t32 < lower_bound) {
                    ++nlow;
                    t32 = lower_bound;
                }
            }
            thr[(size_t)d * C + cell] = t32;
        }
        wave_sync();
    }
    if (lane < nout) {
        if (kmin != 0xFFFFFFFFu) atomicMin(&stats->min_key, kmin);
        if (kmax != 0u) atomicMax(&stats->max_key, kmax);
        if (nlow) atomicAdd(&stats->n_too_low, nlow);
        if (nhigh) atomicAdd(&stats->n_too_high, nhigh);
    }
}

// ------------------------------------------------------------------------------------------------
// K_T (band algorithm, the default): thresholds from per-cell windowed CUMULATIVE level counts.
//
// Workgroup = one tile of TR x TC = 256 grid cells (outputs are the inner (TR-2p) x (TC-2p) cells,
// the rim only feeds the ws x ws pooling) and a block of up to 16 consecutive dayofyears.
// Lane t owns cell t: its LDS column lev[.][t] holds, for the current day, the number of samples of
// the wd-day window of THAT cell with level <= k, for every level k (uint16, two levels per dword;
// the column is private to the lane, so building it needs no atomics and is bank-conflict free).
//   P1  day 0 of the block: count the wd buckets from scratch; later days: undo the prefix sum, add
//       the entering bucket, remove the leaving one; prefix-sum again.
//   P2  every output lane finds the smallest level whose POOLED cumulative count (sum of the 25
//       neighbour columns at that level, integer => exact) exceeds q*total, starting from the
//       previous day's level (2-4 probes of (2p+1)^2 LDS reads instead of scanning all bins).
// Pass 0 uses coarse levels (groups of 2^shift bins) and yields the group holding the quantile bin
// for every (cell, day); the following pass(es) use one level per bin inside the band of groups the
// tile actually needs (<= 64 bins per pass) and produce the exact iu, cs[iu-1], cs[iu] of
// detect.py:2510-2550.  Work per (cell, day) is O(levels + samples entering/leaving), independent
// of the 25-fold spatial fan-out that dominates the sliding-histogram kernel above.
// ------------------------------------------------------------------------------------------------
#define TB_NLP 33
#define TB_LS 34
#define TB_DMAX 32
#define TB_PRE 8
#ifndef TB_BATCH
#define TB_BATCH 16
#endif

template <int P, int TC, int NT>
__global__ void __launch_bounds__(NT)
k_thr_band(const unsigned short* __restrict__ bins, long T_out, long C, int ny, int nx, int row0, int row1, int tiles_x,
           int Dd, int shift, int env_exact, int ablate, const int* __restrict__ doy_start,
           const float* __restrict__ first_anom, const float* __restrict__ centres, int nb, double q, int wd,
           float lower_bound, float upper_bound, float* __restrict__ thr, marex_thr_stats* __restrict__ stats,
           unsigned char* __restrict__ gscratch, int coarse_pd) {
    constexpr int TR = NT / TC;
    constexpr int OR = TR - 2 * P, OC = TC - 2 * P;
    // lane-major level columns: TB_LS dwords (= 68 uint16 levels) per lane.  The stride 34 keeps 8-byte
    // alignment and makes 8-byte accesses of 32 consecutive lanes hit 64 distinct banks.
    __shared__ unsigned lev[NT * TB_LS];
    // per (day of the block, lane) state byte; every thread touches only its own bytes.  256-thread tiles keep it
    // in LDS; 1024-thread tiles (LDS is full of level columns) in a global scratch slab, which lifts the limit on Dd
    __shared__ unsigned char gst_lds[NT > 256 ? 1 : TB_DMAX][NT];
    unsigned char (*gst)[NT] = gst_lds;
    if (NT > 256)
        gst = reinterpret_cast<unsigned char (*)[NT]>(
            gscratch + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * (size_t)Dd * NT);
    __shared__ int s_gmin, s_gmax, s_unres;
    __shared__ unsigned tot_s[NT];  // per tile cell: number of samples in its window (top of the cumulative column)

    const int t = threadIdx.x;
    // lane -> tile cell: rows are rotated by P so that the output rows P .. TR-P-1 fill the FIRST waves completely and
    // the halo rows share the last one(s), which then skip the per-output phase (a 16x16 tile with P = 2: three waves
    // at 48/64 output lanes + one idle, instead of 24+48+48+24 over four)
    const int tc = t % TC;
    const int tr = (TR > 1) ? (t / TC + P) % TR : 0;
    const int ty = (int)blockIdx.x / tiles_x, tx = (int)blockIdx.x - ty * tiles_x;
    const int jt0 = row0 + ty * OR, it0 = tx * OC;
    const int j = (ny > 0) ? jt0 - P + tr : 0;
    const int icol = it0 - P + tc;
    bool cell_valid;
    long cell;
    if (ny > 0) {
        int gi = icol % nx;
        if (gi < 0) gi += nx;
        cell_valid = (j >= 0 && j < ny);
        cell = (long)j * nx + gi;
    } else {
        cell_valid = icol < nx;
        cell = icol;
    }
    const bool is_out = tr >= P && tr < TR - P && tc >= P && tc < TC - P && j < row1 && icol < nx;
    const int d_begin = (int)blockIdx.y * Dd;
    const int ndays = (NDOY - d_begin) < Dd ? (NDOY - d_begin) : Dd;
    const int pd = wd / 2;
    const int ngroups = ((nb - 1) >> shift) + 1;
    const int gsz = 1 << shift;   // bins per coarse group
    const int gpp = 64 >> shift;  // groups per fine pass (>= 1): band of gpp << shift <= 64 bins
    bool land = true;
    if (is_out) land = !(first_anom[cell] == first_anom[cell]);
    if (!__syncthreads_or(is_out && !land)) {  // nothing but land in this tile: all thresholds NaN
        if (is_out)
            for (int dd = 0; dd < ndays; ++dd) thr[(size_t)(d_begin + dd) * C + cell] = nan_f();
        return;
    }

    unsigned* mycol = &lev[(tr * TC + tc) * TB_LS];  // columns are indexed by tile cell: neighbour offsets stay linear
    uint2* mycol2 = reinterpret_cast<uint2*>(mycol);
    for (int r = 0; r < TB_LS / 2; ++r) mycol2[r] = make_uint2(0u, 0u);
    if (t == 0) {
        s_gmin = 255;
        s_gmax = -1;
        s_unres = env_exact;
    }

    // level mapping of the current pass: level(b) = clamp((b >> lsh) + loff, 0, lhi)
    //   coarse: lsh = shift, loff = 0, lhi = ngroups - 1;  fine band B0..B0+BW-1: lsh = 0, loff = 1 - B0, lhi = BW + 1
    bool fine = false;
    int B0 = 0, BW = 0, nlev = ngroups;
    int lsh = shift, loff = 0, lhi = ngroups - 1;
    const unsigned short* colbase = bins + bins_index(0, cell_valid ? cell : 0, T_out);  // rows are 16 elements apart
    // Lanes outside the grid stream cell 0 (valid memory, uniform loop bounds); their bumps are masked off.
    // first TB_PRE samples of a dayofyear bucket of this lane's cell, kept in registers
    struct Pre {
        int b[TB_PRE];
        int r0, nd;
    };
    auto load_bucket = [&](int d0) {
        Pre pr;
        pr.r0 = doy_start[d0];
        pr.nd = doy_start[d0 + 1] - pr.r0;  // uniform
        const unsigned short* col = colbase + (size_t)pr.r0 * 16;
#pragma unroll
        for (int u = 0; u < TB_PRE; ++u) pr.b[u] = (u < pr.nd) ? (int)col[(size_t)u * 16] : nb;
        return pr;
    };
    // One sample: +-1 on its level of the lane's packed uint16 column -- 8 vector instructions + 1 LDS atomic:
    //   k = med3((b >> lsh) + loff, 0, lhi); odd = k & 1; byte offset in the column = 2 * (k - odd);
    //   value = +-(1 << 16 * odd) as a 24-bit multiply-add; NaN-bin samples (b == nb) add zero.
    typedef __attribute__((address_space(3))) unsigned lds_u32;
    const unsigned col_lds = (unsigned)(size_t)(lds_u32*)mycol;  // 32-bit LDS byte address of the lane's column
    const int mul_p = 65535, mul_n = -65535, one_p = 1, one_n = -1;
    auto bump = [&](int b, int sgn) {
        int k = (b >> lsh) + loff;
        asm("v_med3_i32 %0, %1, 0, %2" : "=v"(k) : "v"(k), "v"(lhi));
        const int odd = k & 1;
        const int even = k & ~1;
        unsigned addr;
        asm("v_lshl_add_u32 %0, %1, 1, %2" : "=v"(addr) : "v"(even), "v"(col_lds));
        int v;
        asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(v) : "v"(odd), "v"(sgn > 0 ? mul_p : mul_n), "v"(sgn > 0 ? one_p : one_n));
        v = (b < nb) ? v : 0;
        __hip_atomic_fetch_add((lds_u32*)(size_t)addr, (unsigned)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    };
    auto apply_bucket = [&](const Pre& pr, int sgn) {
        if (cell_valid) {
#pragma unroll
            for (int u = 0; u < TB_PRE; ++u) bump(pr.b[u], sgn);
        }
        if (pr.nd > TB_PRE) {  // long buckets (many years): stream the rest, TB_BATCH loads in flight
            const unsigned short* col = colbase + (size_t)pr.r0 * 16;
            int r = TB_PRE;
            for (; r + TB_BATCH <= pr.nd; r += TB_BATCH) {
                int bb[TB_BATCH];
#pragma unroll
                for (int u = 0; u < TB_BATCH; ++u) bb[u] = (int)col[(size_t)(r + u) * 16];
                if (cell_valid) {
#pragma unroll
                    for (int u = 0; u < TB_BATCH; ++u) bump(bb[u], sgn);
                }
            }
            if (r < pr.nd) {  // last, partial batch in ONE round trip: clamped row index, surplus samples add zero
                int bb[TB_BATCH];
                const int last = pr.nd - 1;
#pragma unroll
                for (int u = 0; u < TB_BATCH; ++u) {
                    const int rr = r + u < last ? r + u : last;  // uniform
                    bb[u] = (int)col[(size_t)rr * 16];
                }
                if (cell_valid) {
#pragma unroll
                    for (int u = 0; u < TB_BATCH; ++u) bump(r + u <= last ? bb[u] : nb, sgn);
                }
            }
        }
    };
    // in-place inclusive prefix sum / its inverse over the lane's column, four levels per 8-byte access
    auto prefix = [&](int nlp) -> unsigned {
        unsigned run = 0;
        const int n2 = (nlp + 1) >> 1;
#pragma unroll 4
        for (int i = 0; i < n2; ++i) {
            const uint2 w = mycol2[i];
            const unsigned a0 = (w.x & 0xFFFFu) + run, a1 = (w.x >> 16) + a0;
            const unsigned a2 = (w.y & 0xFFFFu) + a1, a3 = (w.y >> 16) + a2;
            run = a3;
            mycol2[i] = make_uint2(a0 | (a1 << 16), a2 | (a3 << 16));
        }
        return run;  // all samples of the window (levels past the last used one are empty)
    };
    auto unprefix = [&](int nlp) {
        unsigned prev = 0;
        const int n2 = (nlp + 1) >> 1;
#pragma unroll 4
        for (int i = 0; i < n2; ++i) {
            const uint2 w = mycol2[i];
            const unsigned a0 = w.x & 0xFFFFu, a1 = w.x >> 16, a2 = w.y & 0xFFFFu, a3 = w.y >> 16;
            mycol2[i] = make_uint2((a0 - prev) | ((a1 - a0) << 16), (a2 - a1) | ((a3 - a2) << 16));
            prev = a3;
        }
    };
    // pooled cumulative count at ONE level k of this lane's (2P+1)^2 neighbourhood
    auto pooled = [&](int k) {
        const unsigned* base = mycol + (k >> 1);
        const int sh16 = (k & 1) * 16;
        int sum = 0;
#pragma unroll
        for (int dr = -P; dr <= P; ++dr)
#pragma unroll
            for (int dc = -P; dc <= P; ++dc) sum += (int)((base[(dr * TC + dc) * TB_LS] >> sh16) & 0xFFFFu);
        return sum;
    };
    // number of samples in the pooled window: the per-cell totals published by the column pass
    auto pooled_tot = [&]() {
        const unsigned* base = &tot_s[tr * TC + tc];
        int sum = 0;
#pragma unroll
        for (int dr = -P; dr <= P; ++dr)
#pragma unroll
            for (int dc = -P; dc <= P; ++dc) sum += (int)base[dr * TC + dc];
        return sum;
    };
    // pooled cumulative counts at the EIGHT levels start .. start+7 (start % 4 == 0, start <= 60): two 8-byte
    // reads per neighbour, packed 16-bit adds (the host guarantees pooled counts < 65536)
    auto window = [&](int start, int (&Wv)[8]) {
        const unsigned* base = mycol + (start >> 1);
        unsigned a0 = 0, a1 = 0, a2 = 0, a3 = 0;
#pragma unroll
        for (int dr = -P; dr <= P; ++dr)
#pragma unroll
            for (int dc = -P; dc <= P; ++dc) {
                const uint2* p2 = reinterpret_cast<const uint2*>(base + (dr * TC + dc) * TB_LS);
                const uint2 u = p2[0], v = p2[1];
                a0 += u.x;
                a1 += u.y;
                a2 += v.x;
                a3 += v.y;
            }
        Wv[0] = (int)(a0 & 0xFFFFu);
        Wv[1] = (int)(a0 >> 16);
        Wv[2] = (int)(a1 & 0xFFFFu);
        Wv[3] = (int)(a1 >> 16);
        Wv[4] = (int)(a2 & 0xFFFFu);
        Wv[5] = (int)(a2 >> 16);
        Wv[6] = (int)(a3 & 0xFFFFu);
        Wv[7] = (int)(a3 >> 16);
    };
    // Smallest level k < khi whose pooled cumulative count exceeds qpos (khi if none); ck = that count,
    // cb = the count at k-1 (0 for k == 0).  Counts are integers, so "count <= qpos" is the integer test
    // "count <= floor(qpos)".  The 8-level window starts two levels below the hint (previous day's level)
    // and slides by four until it brackets the answer -- one pass in the common case.
    auto find_level = [&](int hint, int klo, int khi, double qpos, bool /*counts*/, int& ck, int& cb) {
        const int qf = (int)floor(qpos);
        int top = (khi - 1) & ~3;  // last useful window start
        if (top > 60) top = 60;
        if (top < 0) top = 0;
        int start = ((hint >= 0 ? hint : ((klo + khi) >> 1)) - 2) & ~3;
        start = start < 0 ? 0 : (start > top ? top : start);
        ck = 0;
        cb = 0;
        for (;;) {
            int Wv[8];
            window(start, Wv);
            const int m = (khi - start) < 8 ? (khi - start) : 8;  // levels >= khi do not exist
            int n = 0;
#pragma unroll
            for (int i = 0; i < 8; ++i) n += (i < m) && (Wv[i] <= qf);
            if (n == 0) {
                if (start == 0) {
                    ck = Wv[0];
                    return 0;
                }
                start -= 4;  // the answer is at or below `start`: bring level start-1 into view
                continue;
            }
            if (n == m) {
                if (m < 8 || start >= top) {  // no existing level exceeds qpos
                    cb = Wv[m - 1];
                    return khi;
                }
                start += 4;
                continue;
            }
            cb = 0;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (i == n - 1) cb = Wv[i];
                if (i == n) ck = Wv[i];
            }
            return start + n;
        }
    };

    unsigned kmin = 0xFFFFFFFFu, kmax = 0u, nlow = 0u, nhigh = 0u;
    // gst[day][lane]: 255 = nothing (left) to do, 254 = quantile group unknown, 0..31 = coarse group known
    for (int dd = 0; dd < ndays; ++dd) gst[dd][t] = (is_out && !land) ? 254 : 255;
    if (is_out && land)
        for (int dd = 0; dd < ndays; ++dd) thr[(size_t)(d_begin + dd) * C + cell] = nan_f();  // detect.py:2704
    int g_base = 0;

    // exact threshold of one output-day from level k of the current band (ck = cs[iu], cb = cs[iu-1])
    auto emit_threshold = [&](int d, int iu, int ck, int cb, double qpos) {
        const int il = iu > 0 ? iu - 1 : 0;
        const int cs_iu = ck;
        const int cs_il = iu > 0 ? cb : ck;
        const int diff = cs_iu - cs_il;
        const double frac = diff > 0 ? (qpos - (double)cs_il) / (double)diff : 0.5;
        const float dc = centres[iu] - centres[il];
        const double prod = frac * (double)dc;
        float t32 = (float)((double)centres[il] + prod);
        if (iu == 0) t32 = centres[0];
        const unsigned key = ordered_key(t32);
        kmin = key < kmin ? key : kmin;
        kmax = key > kmax ? key : kmax;
        if (t32 > upper_bound) ++nhigh;
        if (t32 < lower_bound) {
            ++nlow;
            t32 = lower_bound;
        }
        thr[(size_t)d * C + cell] = t32;
    };

    // One sweep over the first nd_pass days of the block.
    //   mode 0  coarse levels: group of the quantile bin for every output-day still marked 254
    //   mode 1  fine levels, SPECULATIVE band (chosen from day 0): resolve what falls inside the band,
    //           flag the rest (254) for the exact path
    //   mode 2  fine levels, exact band g_base..: resolve the output-days whose group lies in the band
    // init_pd < pd: day 0 of a coarse sweep sees only the 2*init_pd+1 central buckets -- good enough to PLACE the
    // speculative band (a wrong guess only sends the block to the exact path), never used for a result
    auto sweep = [&](int mode, int nd_pass, int ng, int init_pd) {
        if (mode == 0) {
            fine = false;
            nlev = ngroups;
            lsh = shift;
            loff = 0;
            lhi = ngroups - 1;
        } else {
            fine = true;
            B0 = g_base << shift;
            BW = ng << shift;
            if (B0 + BW > nb) BW = nb - B0;
            nlev = BW + 2;
            lsh = 0;
            loff = 1 - B0;
            lhi = BW + 1;
        }
        const int nlp = (nlev + 1) >> 1;
        int hint = -1;
        Pre pin, pout;  // entering / leaving bucket of the NEXT day, prefetched across the barrier
        for (int dd = 0; dd < nd_pass; ++dd) {
            const int d = d_begin + dd;
            // ---------------- P1: this lane's column
            if (dd == 0) {
                for (int r = 0; r < TB_LS / 2; ++r) mycol2[r] = make_uint2(0u, 0u);
                if (!(ablate & 4)) {
                    Pre cur = load_bucket(((d - init_pd) % NDOY + NDOY) % NDOY);
                    for (int o = -init_pd + 1; o <= init_pd; ++o) {
                        const Pre nxt = load_bucket(((d + o) % NDOY + NDOY) % NDOY);
                        apply_bucket(cur, +1);
                        cur = nxt;
                    }
                    apply_bucket(cur, +1);
                }
            } else {
                if (!(ablate & 2)) unprefix(nlp);
                if (!(ablate & 4)) {
                    apply_bucket(pin, +1);
                    apply_bucket(pout, -1);
                }
            }
            if (!(ablate & 2)) tot_s[tr * TC + tc] = prefix(nlp);
            if (dd + 1 < nd_pass && !(ablate & 4)) {
                pin = load_bucket((d + 1 + pd) % NDOY);
                pout = load_bucket(((d - pd) % NDOY + NDOY) % NDOY);
            }
            __syncthreads();
            // ---------------- P2: quantile level of this lane's output cell
            const int g = (ablate & 1) ? 255 : gst[dd][t];
            if (mode == 0) {
                if (g == 254) {
                    const int tot = pooled_tot();
                    if (tot > 0) {
                        int ck, cb;
                        int gg = find_level(hint, 0, nlev, q * (double)tot, false, ck, cb);
                        if (gg >= nlev) gg = nlev - 1;  // nothing above qpos: iu clips to nb-1
                        hint = gg;
                        gst[dd][t] = (unsigned char)gg;
                        atomicMin(&s_gmin, gg);
                        atomicMax(&s_gmax, gg);
                    } else if (init_pd == pd) {
                        gst[dd][t] = 255;
                        thr[(size_t)d * C + cell] = nan_f();  // empty window
                    }
                }
            } else if (mode == 1) {
                if (g != 255) {
                    const int tot = pooled_tot();
                    if (tot > 0) {
                        const double qpos = q * (double)tot;
                        if (hint < 0 && g < 254) hint = ((g - g_base) << shift) + 1 + (gsz >> 1);
                        int ck, cb;
                        const int k = find_level(hint, 1, BW + 1, qpos, true, ck, cb);
                        // inside the band iff cs[B0-1] <= qpos (k == 1 needs the check) and some band bin exceeds qpos
                        const bool ok = k >= 1 && k <= BW;
                        if (ok) {
                            hint = k;
                            emit_threshold(d, B0 + k - 1, ck, cb, qpos);
                            gst[dd][t] = 255;
                        } else {
                            hint = -1;
                            gst[dd][t] = 254;
                            s_unres = 1;
                        }
                    } else {
                        gst[dd][t] = 255;
                        thr[(size_t)d * C + cell] = nan_f();
                    }
                }
            } else {
                if (g < 254 && g >= g_base && g < g_base + ng) {
                    const int tot = pooled_tot();
                    const double qpos = q * (double)tot;
                    // the quantile bin lies inside group g: levels klo .. khi-1 of this band
                    const int klo = ((g - g_base) << shift) + 1;
                    int khi = klo + gsz;
                    if (khi > BW + 1) khi = BW + 1;
                    int ck, cb;
                    int k = find_level(hint, klo, khi, qpos, true, ck, cb);
                    int iu = B0 + k - 1;
                    if (k >= khi) {  // no bin exceeds qpos (q == 1): searchsorted gives nb, clipped to nb-1
                        iu = nb - 1;
                        k = iu - B0 + 1;
                        ck = pooled(k);
                        cb = pooled(k - 1);
                    }
                    hint = k;
                    emit_threshold(d, iu, ck, cb, qpos);
                    gst[dd][t] = 255;
                } else {
                    hint = -1;
                }
            }
            __syncthreads();
        }
    };

    __syncthreads();
    sweep(0, 1, 0, (coarse_pd >= 0 && coarse_pd < pd) ? coarse_pd : pd);  // coarse, day 0 only
    int gmin = s_gmin, gmax = s_gmax;
    __syncthreads();
    if (!env_exact && gmax >= 0 && gmax - gmin + 1 <= gpp) {
        // speculative band of gpp groups placed around what day 0 needs (room for drift on both sides)
        const int spare = gpp - (gmax - gmin + 1);
        g_base = gmin - (spare + 1) / 2;
        if (g_base < 0) g_base = 0;
        if (g_base + gpp > ngroups) g_base = ngroups - gpp > 0 ? ngroups - gpp : 0;
        sweep(1, ndays, gpp < ngroups ? gpp : ngroups, pd);
    } else if (t == 0) {
        s_unres = 1;
    }
    __syncthreads();
    if (s_unres && !(ablate & 8)) {  // exact path for whatever is not resolved yet
        for (int dd = 0; dd < ndays; ++dd)
            if (gst[dd][t] < 254) gst[dd][t] = 254;  // day-0 groups of a skipped speculative sweep: redo
        __syncthreads();
        if (t == 0) {
            s_gmin = 255;
            s_gmax = -1;
        }
        __syncthreads();
        sweep(0, ndays, 0, pd);
        gmin = s_gmin;
        gmax = s_gmax;
        for (g_base = gmin; g_base <= gmax; g_base += gpp) {
            const int ng = (gmax - g_base + 1) < gpp ? (gmax - g_base + 1) : gpp;
            sweep(2, ndays, ng, pd);
        }
    }
    // statistics: wave reduction, one set of global atomics per wave
    for (int sft = 32; sft > 0; sft >>= 1) {
        const unsigned a = __shfl_down(kmin, sft, 64), b = __shfl_down(kmax, sft, 64);
        kmin = a < kmin ? a : kmin;
        kmax = b > kmax ? b : kmax;
        nlow += __shfl_down(nlow, sft, 64);
        nhigh += __shfl_down(nhigh, sft, 64);
    }
    if ((t & 63) == 0) {
        if (kmin != 0xFFFFFFFFu) atomicMin(&stats->min_key, kmin);
        if (kmax != 0u) atomicMax(&stats->max_key, kmax);
        if (nlow) atomicAdd(&stats->n_too_low, nlow);
        if (nhigh) atomicAdd(&stats->n_too_high, nhigh);
    }
}

extern "C" int marex_hobday_thresholds_f32(marex_ctx* ctx, const uint16_t* bins, int64_t T_out, int64_t C, int ny,
                                           int nx, const int32_t* doy_start, int max_bucket,
                                           const float* first_anom, const float* centres, int nb, double q, int wd,
                                           int ws, float lower_bound, float upper_bound, int row0, int row1,
                                           float* thr_doy_major, marex_thr_stats* stats) {
    if (!ctx) return -1;
    if (!bins || !doy_start || !first_anom || !centres || !thr_doy_major || !stats || T_out <= 0 || C <= 0)
        return fail(ctx, -1, "marex_hobday_thresholds_f32: null pointer or empty shape");
    if (wd < 3 || wd > 365 || (wd & 1) == 0)
        return fail(ctx, -1, "marex_hobday_thresholds_f32: window_days_hobday must be odd and in 3..365");
    if (ws < 1 || (ws & 1) == 0) return fail(ctx, -1, "marex_hobday_thresholds_f32: window_spatial_hobday must be odd");
    if (!(q > 0.0 && q <= 1.0)) return fail(ctx, -1, "marex_hobday_thresholds_f32: q must be in (0, 1]");
    if (ny == 0) {
        if (ws > 1) return fail(ctx, -1, "marex_hobday_thresholds_f32: spatial pooling needs a structured grid");
        nx = (int)C;
        row0 = 0;
        row1 = 1;
    } else if ((int64_t)ny * nx != C) {
        return fail(ctx, -1, "marex_hobday_thresholds_f32: ny*nx != C");
    } else if (row0 < 0 || row1 > ny || row0 >= row1) {
        return fail(ctx, -1, "marex_hobday_thresholds_f32: need 0 <= row0 < row1 <= ny");
    }
    if (nb < 4 || nb > 36000) return fail(ctx, -4, "marex_hobday_thresholds_f32: nb must be in 4..36000");
    HIP_TRY(ctx, hipSetDevice(ctx->device));

    // ---- band algorithm (default) whenever its uint16 level counters and 64-bin bands suffice
    int shift = 0;
    while ((((nb - 1) >> shift) + 1) > 32) ++shift;  // at most 32 coarse groups
    const int algo = env_int("MAREX_THR_ALGO", 0);   // 0 auto, 1 force sliding histograms
    const int p = ws / 2;
    const bool band_ok = (1 << shift) <= 64 && p <= 3 && max_bucket > 0 && (int64_t)max_bucket * wd * ws * ws <= 65535;
    if (algo != 1 && band_ok) {
        // tile: 16x16 cells / 256 threads, or 32x32 / 1024 threads (less halo redundancy, more output lanes)
        // long dayofyear buckets (many years) make the kernel sample-streaming bound: the big tile re-streams
        // 1.31x instead of 1.78x halo cells per output cell (measured 17.4 vs 23.3 ms on an 85-year band)
        const int tile_pref = env_int("MAREX_THR_TILE", max_bucket >= 24 ? 32 : 16);
        // tile_pref 32: 32x32 cells / 1024 threads (one workgroup per CU); 3216: 32 wide x 16 tall / 512 threads (two
        // independent workgroups per CU); 16: 16x16 / 256 threads
        const bool big = (ny > 0 && p > 0) && (tile_pref == 32 || tile_pref == 3216) && (row1 - row0) >= 16 && nx >= 16;
        const bool half = big && tile_pref == 3216;
        const int NT = big ? (half ? 512 : 1024) : 256;
        const int TR = (ny > 0 && p > 0) ? (big ? (half ? 16 : 32) : 16) : 1, TC = NT / TR;
        const int OR = TR - 2 * p, OC = TC - 2 * p;
        int Dd = env_int("MAREX_THR_DD", big ? 48 : TB_DMAX);
        if (Dd < 1 || Dd > (big ? 128 : TB_DMAX)) Dd = big ? 48 : TB_DMAX;
        const int tiles_x = (nx + OC - 1) / OC, tiles_y = (row1 - row0 + OR - 1) / OR;
        dim3 grid((unsigned)(tiles_x * tiles_y), (unsigned)((NDOY + Dd - 1) / Dd));
        unsigned char* gscratch = nullptr;
        if (big) {  // state bytes of the 1024-thread tiles
            const size_t need = (size_t)grid.x * grid.y * (size_t)Dd * NT;
            if (need > ctx->thr_scratch_bytes) {
                if (ctx->thr_scratch) HIP_TRY(ctx, hipFree(ctx->thr_scratch));
                ctx->thr_scratch = nullptr;
                ctx->thr_scratch_bytes = 0;
                HIP_TRY(ctx, hipMalloc((void**)&ctx->thr_scratch, need));
                ctx->thr_scratch_bytes = need;
            }
            gscratch = ctx->thr_scratch;
        }
        const int coarse_pd = env_int("MAREX_THR_COARSE_PD", 1);
#define MAREX_BAND_ARGS bins, (long)T_out, (long)C, ny, nx, row0, row1, tiles_x, Dd, shift, env_int("MAREX_THR_EXACT_PATH", 0), env_int("MAREX_THR_ABLATE", 0), doy_start, first_anom, centres, nb, q, wd, lower_bound, upper_bound, thr_doy_major, stats, gscratch, coarse_pd
        {
            LaunchTimer lt(ctx, MAREX_K_THRESHOLDS);
            if (TR == 1)
                hipLaunchKernelGGL((k_thr_band<0, 256, 256>), grid, dim3(256), 0, ctx->stream, MAREX_BAND_ARGS);
            else if (half && p == 2)
                hipLaunchKernelGGL((k_thr_band<2, 32, 512>), grid, dim3(512), 0, ctx->stream, MAREX_BAND_ARGS);
            else if (big && p == 1)
                hipLaunchKernelGGL((k_thr_band<1, 32, 1024>), grid, dim3(1024), 0, ctx->stream, MAREX_BAND_ARGS);
            else if (big && p == 2)
                hipLaunchKernelGGL((k_thr_band<2, 32, 1024>), grid, dim3(1024), 0, ctx->stream, MAREX_BAND_ARGS);
            else if (big)
                hipLaunchKernelGGL((k_thr_band<3, 32, 1024>), grid, dim3(1024), 0, ctx->stream, MAREX_BAND_ARGS);
            else if (p == 1)
                hipLaunchKernelGGL((k_thr_band<1, 16, 256>), grid, dim3(256), 0, ctx->stream, MAREX_BAND_ARGS);
            else if (p == 2)
                hipLaunchKernelGGL((k_thr_band<2, 16, 256>), grid, dim3(256), 0, ctx->stream, MAREX_BAND_ARGS);
            else
                hipLaunchKernelGGL((k_thr_band<3, 16, 256>), grid, dim3(256), 0, ctx->stream, MAREX_BAND_ARGS);
        }
#undef MAREX_BAND_ARGS
        HIP_TRY(ctx, hipGetLastError());
        return 0;
    }

    // ---- sliding pooled histograms (any nb / ws / bucket size)
    // uint16 counters are enough when even "all samples of the pooled window in one bin" fits
    const bool pack = max_bucket > 0 && (int64_t)max_bucket * wd * ws * ws <= 65535 && !env_int("MAREX_THR_U32", 0);
    const int nbw = pack ? (nb + 1) / 2 : nb;
    int NW = env_int("MAREX_THR_NW", 16);
    if (NW < 1 || NW > 64) NW = 16;
    const size_t budget = 80 * 1024;  // per workgroup of 4 waves: two workgroups per CU
    while (NW > 1 && 4 * (size_t)NW * (nbw + 3) * 4 > budget) NW >>= 1;
    if (NW > nx) NW = nx;
    const size_t lds = 4 * (size_t)NW * (nbw + 3) * 4;
    if (lds > 80 * 1024) return fail(ctx, -4, "marex_hobday_thresholds_f32: %d bins need more than 80 KiB of LDS", nb);
    const int nseg_per_row = (nx + NW - 1) / NW;
    const long nsegs = (long)nseg_per_row * (row1 - row0);
    const unsigned nblocks = (unsigned)((nsegs + 3) / 4);
    auto kern = pack ? k_thresholds<true> : k_thresholds<false>;
    if (lds > 48 * 1024)
        HIP_TRY(ctx, hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    {
        LaunchTimer lt(ctx, MAREX_K_THRESHOLDS);
        hipLaunchKernelGGL(kern, dim3(nblocks), dim3(256), lds, ctx->stream, bins, (long)T_out, (long)C, ny, nx,
                           nseg_per_row, nsegs, NW, doy_start, first_anom, centres, nb, q, wd, ws / 2, lower_bound, upper_bound,
                           row0, thr_doy_major, stats);
    }
    HIP_TRY(ctx, hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------------------------------------
// K_M: extreme[t, c] = anom[t, c] >= thr[doy(t), c]
// One workgroup = (VEC*256 cells, a chunk of consecutive dayofyears).  Rows are visited grouped by
// dayofyear so that one threshold row serves all its timesteps out of registers; 16-byte loads of the
// anomaly row, 4-byte stores of the mask, MASK_UNROLL independent rows in flight per lane.
// ------------------------------------------------------------------------------------------------
#define MASK_DOY_CHUNKS 6
template <int VEC>
__global__ void __launch_bounds__(256)
k_mask_ge(const float* __restrict__ anom, const float* __restrict__ thr, const int* __restrict__ doy_start,
          const int* __restrict__ doy_rows, long C, long c0, long c1, unsigned char* __restrict__ out,
          unsigned long long* __restrict__ n_true) {
    const int nchunk = (int)gridDim.y;  // the dayofyear axis is cut into gridDim.y pieces
    const int dA = (int)blockIdx.y * NDOY / nchunk, dB = ((int)blockIdx.y + 1) * NDOY / nchunk;
    const long c = c0 + ((long)blockIdx.x * 256 + threadIdx.x) * VEC;
    unsigned cnt = 0;
    if (c < c1) {
        for (int d = dA; d < dB; ++d) {
            const int r0 = doy_start[d], r1 = doy_start[d + 1];
            if (VEC == 4) {
                const float4 th = *reinterpret_cast<const float4*>(thr + (size_t)d * C + c);
                int r = r0;
                for (; r + 4 <= r1; r += 4) {
                    size_t off[4];
                    float4 a[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) off[u] = (size_t)doy_rows[r + u] * C + c;
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        typedef float f4_t __attribute__((ext_vector_type(4)));
                        const f4_t q4 = __builtin_nontemporal_load(reinterpret_cast<const f4_t*>(anom + off[u]));
                        a[u] = make_float4(q4.x, q4.y, q4.z, q4.w);
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        uchar4 m;
                        m.x = a[u].x >= th.x;
                        m.y = a[u].y >= th.y;
                        m.z = a[u].z >= th.z;
                        m.w = a[u].w >= th.w;
                        cnt += m.x + m.y + m.z + m.w;
                        __builtin_nontemporal_store((unsigned)m.x | ((unsigned)m.y << 8) | ((unsigned)m.z << 16) | ((unsigned)m.w << 24),
                                                    reinterpret_cast<unsigned*>(out + off[u]));
                    }
                }
                for (; r < r1; ++r) {
                    const size_t off = (size_t)doy_rows[r] * C + c;
                    const float4 a = *reinterpret_cast<const float4*>(anom + off);
                    uchar4 m;
                    m.x = a.x >= th.x;
                    m.y = a.y >= th.y;
                    m.z = a.z >= th.z;
                    m.w = a.w >= th.w;
                    cnt += m.x + m.y + m.z + m.w;
                    *reinterpret_cast<uchar4*>(out + off) = m;
                }
            } else {
                const float th = thr[(size_t)d * C + c];
                for (int r = r0; r < r1; ++r) {
                    const size_t off = (size_t)doy_rows[r] * C + c;
                    const unsigned char m = anom[off] >= th;
                    cnt += m;
                    out[off] = m;
                }
            }
        }
    }
    if (n_true) {
        for (int s = 32; s > 0; s >>= 1) cnt += __shfl_down(cnt, s, 64);
        if ((threadIdx.x & 63) == 0 && cnt) atomicAdd(n_true, (unsigned long long)cnt);
    }
}

extern "C" int marex_mask_ge_doy_f32(marex_ctx* ctx, const float* anom, const float* thr_doy_major,
                                     const int32_t* doy_start, const int32_t* doy_rows, int64_t T_out, int64_t C,
                                     int64_t c0, int64_t c1, uint8_t* extreme, unsigned long long* n_true) {
    if (!ctx) return -1;
    if (!anom || !thr_doy_major || !doy_start || !doy_rows || !extreme || T_out <= 0 || C <= 0)
        return fail(ctx, -1, "marex_mask_ge_doy_f32: null pointer or empty shape");
    if (c0 < 0 || c1 > C || c0 >= c1) return fail(ctx, -1, "marex_mask_ge_doy_f32: need 0 <= c0 < c1 <= C");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const int64_t nc = c1 - c0;
    const bool vec = (C % 4 == 0) && (c0 % 4 == 0) && (c1 % 4 == 0) && (((uintptr_t)anom | (uintptr_t)thr_doy_major) % 16 == 0) && ((uintptr_t)extreme % 4 == 0);
    {
        LaunchTimer lt(ctx, MAREX_K_MASK);
        if (vec) {
            // enough workgroups to fill the chip whatever the number of cells (a 100-yr latitude band has 133 cell blocks)
            const unsigned ncb4 = (unsigned)((nc / 4 + 255) / 256);
            unsigned chunks = (4096 + ncb4 - 1) / ncb4;
            chunks = chunks < MASK_DOY_CHUNKS ? MASK_DOY_CHUNKS : (chunks > 61 ? 61 : chunks);
            dim3 grid(ncb4, chunks);
            hipLaunchKernelGGL(k_mask_ge<4>, grid, dim3(256), 0, ctx->stream, anom, thr_doy_major, doy_start, doy_rows,
                               (long)C, (long)c0, (long)c1, extreme, n_true);
        } else {
            dim3 grid((unsigned)((nc + 255) / 256), MASK_DOY_CHUNKS);
            hipLaunchKernelGGL(k_mask_ge<1>, grid, dim3(256), 0, ctx->stream, anom, thr_doy_major, doy_start, doy_rows,
                               (long)C, (long)c0, (long)c1, extreme, n_true);
        }
    }
    HIP_TRY(ctx, hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------------------------------------
// K_F: fixed-baseline anomaly (detect.py:2299-2397).  Work item = (256 cells, one dayofyear): the
// float32 nanmean of all timesteps of that dayofyear (optionally only reference-period years) in
// ascending time, then anom = x - clim for the same rows (second read comes from L2).  Also emits the
// dayofyear-sorted bins, the t=0 mask and the validation counts like the shifting-baseline kernel.
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
k_fixed_baseline(const float* __restrict__ x, long T, long C, const int* __restrict__ doy_start,
                 const int* __restrict__ doy_rows, const unsigned char* __restrict__ use_row,
                 const float* __restrict__ edges, int nb, float* __restrict__ out,
                 unsigned short* __restrict__ bins, unsigned char* __restrict__ mask,
                 int* __restrict__ invalid_count) {
    extern __shared__ float e[];
    const int d = blockIdx.y;
    const long c = (long)blockIdx.x * 256 + threadIdx.x;
    const bool do_bins = bins != nullptr;
    if (do_bins) {
        for (int i = threadIdx.x; i <= nb; i += 256) e[i] = edges[i];
        __syncthreads();
    }
    if (c >= C) return;
    const float inv_width = do_bins ? (float)(nb - 1) / (e[nb] - e[1]) : 0.f;
    if (d == 0 && mask) mask[c] = finite_f(x[c]) ? 1 : 0;
    const int r0 = doy_start[d], r1 = doy_start[d + 1];
    float acc = 0.f;
    int n = 0, n_invalid = 0;
    for (int r = r0; r < r1; ++r) {
        const int t = doy_rows[r];
        const float v = x[(size_t)t * C + c];
        if (!finite_f(v)) ++n_invalid;
        if ((!use_row || use_row[t]) && v == v) {
            acc += v;
            ++n;
        }
    }
    const float clim = acc / (float)n;  // n == 0 -> NaN
    for (int r = r0; r < r1; ++r) {
        const int t = doy_rows[r];
        const float a = x[(size_t)t * C + c] - clim;
        out[(size_t)t * C + c] = a;
        if (do_bins) bins[bins_index(r, c, T)] = (unsigned short)digitize_bin(a, e, nb, inv_width);
    }
    if (invalid_count && n_invalid) atomicAdd(&invalid_count[c], n_invalid);
}

extern "C" int marex_fixed_baseline_f32(marex_ctx* ctx, const float* x, int64_t T, int64_t C,
                                        const int32_t* doy_start, const int32_t* doy_rows,
                                        const uint8_t* use_row, const float* edges, int nb, float* out,
                                        uint16_t* bins, uint8_t* mask, int32_t* invalid_count) {
    if (!ctx) return -1;
    if (!x || !doy_start || !doy_rows || !out || T <= 0 || C <= 0)
        return fail(ctx, -1, "marex_fixed_baseline_f32: null pointer or empty shape");
    if (bins && (!edges || nb < 4 || nb > 36000)) return fail(ctx, -1, "marex_fixed_baseline_f32: binning needs edges and 4 <= nb <= 36000");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    dim3 grid((unsigned)((C + 255) / 256), NDOY);
    const size_t lds = bins ? ((size_t)nb + 1) * sizeof(float) : 0;
    if (lds > 48 * 1024)
        HIP_TRY(ctx, hipFuncSetAttribute((const void*)k_fixed_baseline, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    {
        LaunchTimer lt(ctx, MAREX_K_FIXED);
        hipLaunchKernelGGL(k_fixed_baseline, grid, dim3(256), lds, ctx->stream, x, (long)T, (long)C, doy_start, doy_rows,
                           use_row, edges, nb, out, bins, mask, invalid_count);
    }
    HIP_TRY(ctx, hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------------------------------------
// stand-alone binning of an anomaly field into the dayofyear-sorted bin matrix (detect.py:2622-2631)
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
k_digitize(const float* __restrict__ anom, long T, long C, const int* __restrict__ rowb_index,
           const float* __restrict__ edges, int nb, int rows_per_block, long T_out, unsigned short* __restrict__ bins) {
    extern __shared__ float e[];
    for (int i = threadIdx.x; i <= nb; i += 256) e[i] = edges[i];
    __syncthreads();
    const long c = (long)blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    const float inv_width = (float)(nb - 1) / (e[nb] - e[1]);
    const long t0 = (long)blockIdx.y * rows_per_block;
    const long t1 = t0 + rows_per_block < T ? t0 + rows_per_block : T;
    for (long t = t0; t < t1; ++t) {
        const int rb = rowb_index[t];
        if (rb >= 0) bins[bins_index(rb, c, T_out)] = (unsigned short)digitize_bin(anom[(size_t)t * C + c], e, nb, inv_width);
    }
}

extern "C" int marex_digitize_f32(marex_ctx* ctx, const float* anom, int64_t T, int64_t C, const int32_t* rowb_index,
                                  const float* edges, int nb, int64_t T_out, uint16_t* bins) {
    if (!ctx) return -1;
    if (!anom || !rowb_index || !edges || !bins || T <= 0 || C <= 0 || nb < 4 || nb > 36000 || T_out <= 0)
        return fail(ctx, -1, "marex_digitize_f32: bad argument");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const int rows = 32;
    dim3 grid((unsigned)((C + 255) / 256), (unsigned)((T + rows - 1) / rows));
    const size_t lds = ((size_t)nb + 1) * sizeof(float);
    if (lds > 48 * 1024)
        HIP_TRY(ctx, hipFuncSetAttribute((const void*)k_digitize, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    {
        LaunchTimer lt(ctx, MAREX_K_FIXED);
        hipLaunchKernelGGL(k_digitize, grid, dim3(256), lds, ctx->stream, anom, (long)T, (long)C, rowb_index, edges, nb,
                           rows, (long)T_out, bins);
    }
    HIP_TRY(ctx, hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------------------------------------
// K_D: polynomial / harmonic detrend (detect.py:2143-2224).  One lane per cell streams its series:
//   pass 1  coef[k] = sum_t pmodel[t][k] * x[t]      (float64, ascending t, separately rounded mul/add)
//   pass 2  resid[t] = x[t] - fl32( sum_k model[k][t] * coef[k] )   and the float64 sum of resid
//   pass 3  (force_zero_mean) resid[t] -= fl32( sum / T )
// n_coef <= 8 (1 + polynomial orders + 4 harmonics): 4 flop per byte, far below any MFMA use.
// The model tables are tiny ([T, n_coef] float64) and read through the scalar cache (uniform address).
// ------------------------------------------------------------------------------------------------
#define DETREND_MAXC 12
#define DETREND_TBLOCK 1024  // timesteps per partial sum (arithmetic contract, oracle.DETREND_TBLOCK)

// Reductions over time are split into blocks of DETREND_TBLOCK timesteps so that the grid is (cell blocks x time
// blocks) instead of one thread walking 36 500 rows: float64 partial sums per block in ascending t, combined in
// ascending block order -- a fixed order, mirrored by the oracle.
__global__ void __launch_bounds__(256)
k_detrend_partial(const float* __restrict__ x, long T, long C, const double* __restrict__ pmodel /*[T][n]*/, int n_coef,
                  double* __restrict__ partial /*[ntb][n][C]*/, int* __restrict__ invalid_count) {
    const long c = (long)blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    const long t0 = (long)blockIdx.y * DETREND_TBLOCK;
    const long t1 = t0 + DETREND_TBLOCK < T ? t0 + DETREND_TBLOCK : T;
    double acc[DETREND_MAXC];
#pragma unroll
    for (int k = 0; k < DETREND_MAXC; ++k) acc[k] = 0.0;
    int n_invalid = 0;
#pragma unroll 4
    for (long t = t0; t < t1; ++t) {
        const float v = x[(size_t)t * C + c];
        n_invalid += finite_f(v) ? 0 : 1;
        const double vd = (double)v;
        const double* pm = pmodel + (size_t)t * n_coef;
#pragma unroll
        for (int k = 0; k < DETREND_MAXC; ++k)
            if (k < n_coef) acc[k] += pm[k] * vd;
    }
    for (int k = 0; k < n_coef; ++k) partial[((size_t)blockIdx.y * n_coef + k) * C + c] = acc[k];
    if (invalid_count && n_invalid) atomicAdd(&invalid_count[c], n_invalid);
}

__global__ void __launch_bounds__(256)
k_detrend_combine(const float* __restrict__ x, long C, int ntb, int n_coef, const double* __restrict__ partial,
                  double* __restrict__ coef /*[n][C]*/, unsigned char* __restrict__ mask) {
    const long c = (long)blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    for (int k = 0; k < n_coef; ++k) {
        double s = 0.0;
        for (int b = 0; b < ntb; ++b) s += partial[((size_t)b * n_coef + k) * C + c];
        coef[(size_t)k * C + c] = s;
    }
    if (mask) mask[c] = finite_f(x[c]) ? 1 : 0;
}

__global__ void __launch_bounds__(256)
k_detrend_resid(const float* __restrict__ x, long T, long C, const double* __restrict__ model_t /*[T][n]*/, int n_coef,
                const double* __restrict__ coef, float* __restrict__ out, double* __restrict__ psum /*[ntb][C]*/) {
    const long c = (long)blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    const long t0 = (long)blockIdx.y * DETREND_TBLOCK;
    const long t1 = t0 + DETREND_TBLOCK < T ? t0 + DETREND_TBLOCK : T;
    double cf[DETREND_MAXC];
#pragma unroll
    for (int k = 0; k < DETREND_MAXC; ++k) cf[k] = k < n_coef ? coef[(size_t)k * C + c] : 0.0;
    double sum = 0.0;
#pragma unroll 4
    for (long t = t0; t < t1; ++t) {
        const double* mt = model_t + (size_t)t * n_coef;
        double trend = 0.0;
#pragma unroll
        for (int k = 0; k < DETREND_MAXC; ++k)
            if (k < n_coef) trend += mt[k] * cf[k];
        const float r = x[(size_t)t * C + c] - (float)trend;
        out[(size_t)t * C + c] = r;
        sum += (double)r;
    }
    psum[(size_t)blockIdx.y * C + c] = sum;
}

__global__ void __launch_bounds__(256)
k_detrend_mean(long T, long C, int ntb, const double* __restrict__ psum, float* __restrict__ mean) {
    const long c = (long)blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    double s = 0.0;
    for (int b = 0; b < ntb; ++b) s += psum[(size_t)b * C + c];
    mean[c] = (float)(s / (double)T);
}

__global__ void __launch_bounds__(256)
k_detrend_sub(long T, long C, const float* __restrict__ mean, float* __restrict__ out) {
    const long c = ((long)blockIdx.x * 256 + threadIdx.x) * 4;
    if (c >= C) return;
    const long t0 = (long)blockIdx.y * 64;
    const long t1 = t0 + 64 < T ? t0 + 64 : T;
    if (c + 4 <= C && (C & 3) == 0) {
        const float4 m = *reinterpret_cast<const float4*>(mean + c);
        for (long t = t0; t < t1; ++t) {
            float4* p = reinterpret_cast<float4*>(out + (size_t)t * C + c);
            float4 v = *p;
            v.x -= m.x;
            v.y -= m.y;
            v.z -= m.z;
            v.w -= m.w;
            *p = v;
        }
    } else {
        for (long t = t0; t < t1; ++t)
            for (long cc = c; cc < C && cc < c + 4; ++cc) out[(size_t)t * C + cc] -= mean[cc];
    }
}

extern "C" int marex_detrend_f32(marex_ctx* ctx, const float* x, int64_t T, int64_t C, const double* pmodel,
                                 const double* model_t, int n_coef, int force_zero_mean, float* out, uint8_t* mask,
                                 int32_t* invalid_count) {
    if (!ctx) return -1;
    if (!x || !pmodel || !model_t || !out || T <= 0 || C <= 0) return fail(ctx, -1, "marex_detrend_f32: null pointer or empty shape");
    if (n_coef < 1 || n_coef > DETREND_MAXC) return fail(ctx, -4, "marex_detrend_f32: n_coef must be in 1..%d", DETREND_MAXC);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const int ntb = (int)((T + DETREND_TBLOCK - 1) / DETREND_TBLOCK);
    // scratch: partial [ntb][n][C] f64 (reused as psum [ntb][C]) + coef [n][C] f64 + mean [C] f32
    const size_t need = ((size_t)ntb * n_coef + n_coef + 1) * (size_t)C * sizeof(double);
    if (need > ctx->detrend_scratch_bytes) {
        if (ctx->detrend_scratch) HIP_TRY(ctx, hipFree(ctx->detrend_scratch));
        ctx->detrend_scratch = nullptr;
        ctx->detrend_scratch_bytes = 0;
        HIP_TRY(ctx, hipMalloc((void**)&ctx->detrend_scratch, need));
        ctx->detrend_scratch_bytes = need;
    }
    double* partial = reinterpret_cast<double*>(ctx->detrend_scratch);
    double* coef = partial + (size_t)ntb * n_coef * C;
    float* mean = reinterpret_cast<float*>(coef + (size_t)n_coef * C);
    if (invalid_count) HIP_TRY(ctx, hipMemsetAsync(invalid_count, 0, (size_t)C * sizeof(int), ctx->stream));
    const unsigned ncb = (unsigned)((C + 255) / 256);
    {
        LaunchTimer lt(ctx, MAREX_K_DETREND);
        hipLaunchKernelGGL(k_detrend_partial, dim3(ncb, ntb), dim3(256), 0, ctx->stream, x, (long)T, (long)C, pmodel, n_coef,
                           partial, invalid_count);
        hipLaunchKernelGGL(k_detrend_combine, dim3(ncb), dim3(256), 0, ctx->stream, x, (long)C, ntb, n_coef, partial, coef,
                           mask);
        hipLaunchKernelGGL(k_detrend_resid, dim3(ncb, ntb), dim3(256), 0, ctx->stream, x, (long)T, (long)C, model_t, n_coef,
                           coef, out, partial);
        if (force_zero_mean) {
            hipLaunchKernelGGL(k_detrend_mean, dim3(ncb), dim3(256), 0, ctx->stream, (long)T, (long)C, ntb, partial, mean);
            hipLaunchKernelGGL(k_detrend_sub, dim3((unsigned)((C / 4 + 256) / 256), (unsigned)((T + 63) / 64)), dim3(256), 0,
                               ctx->stream, (long)T, (long)C, mean, out);
        }
    }
    HIP_TRY(ctx, hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------------------------------------
// K_X: exact Hobday percentile (detect.py:1921-1956): np.nanpercentile over the finite anomalies of the
// wd-day window of every (cell, dayofyear), no spatial pooling.  Only the two order statistics around
// (m-1)*q matter, so each lane (= cell) streams its window once and keeps the K largest keys
// (key = v for q >= 0.5, -v otherwise) in a private, descending LDS column; K = tail size + slack is
// a few percent of the window.  The interpolation mirrors NumPy 2.x float32 "linear" (SURVEY A.8).
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
k_hobday_exact(const float* __restrict__ anom, long C, const int* __restrict__ doy_start,
               const int* __restrict__ doy_rows, int wd, float q32, int upper, int K,
               float* __restrict__ thr, int* __restrict__ overflow) {
    extern __shared__ float topk[];  // [K][blockDim.x]
    const int BT = blockDim.x;
    const int d = blockIdx.y;
    const long c = (long)blockIdx.x * BT + threadIdx.x;
    if (c >= C) return;
    float* col = topk + threadIdx.x;
    const int pd = wd / 2;
    int m = 0, cnt = 0;
    for (int o = -pd; o <= pd; ++o) {
        const int dd = ((d + o) % NDOY + NDOY) % NDOY;
        const int r0 = doy_start[dd], r1 = doy_start[dd + 1];
        for (int r = r0; r < r1; ++r) {
            const float v = anom[(size_t)doy_rows[r] * C + c];
            if (!(v == v)) continue;
            ++m;
            const float key = upper ? v : -v;
            if (cnt < K || key > col[(size_t)(K - 1) * BT]) {
                int i = (cnt < K ? cnt : K - 1) - 1;  // last element that stays
                while (i >= 0 && col[(size_t)i * BT] < key) {
                    col[(size_t)(i + 1) * BT] = col[(size_t)i * BT];
                    --i;
                }
                col[(size_t)(i + 1) * BT] = key;
                if (cnt < K) ++cnt;
            }
        }
    }
    float res = nan_f();
    if (m > 0) {
        const float h = (float)(m - 1) * q32;
        int lo = (int)floorf(h);
        float g = h - (float)lo;
        int hi = lo + 1;
        if (lo >= m - 1) {
            lo = m - 1;
            hi = m - 1;
        }
        // ascending rank r lives at descending index m-1-r (upper) or at index r of the negated keys (lower)
        const int ia = upper ? m - 1 - lo : lo, ib = upper ? m - 1 - hi : hi;
        if (ia >= cnt || ib >= cnt || ia < 0 || ib < 0) {
            atomicAdd(overflow, 1);
        } else {
            float a = col[(size_t)ia * BT], b = col[(size_t)ib * BT];
            if (!upper) {
                a = -a;
                b = -b;
            }
            const float dba = b - a;
            res = a + dba * g;
            if (g >= 0.5f) res = b - dba * (1.0f - g);
        }
    }
    thr[(size_t)d * C + c] = res;
}

extern "C" int marex_hobday_exact_f32(marex_ctx* ctx, const float* anom, int64_t T_out, int64_t C,
                                      const int32_t* doy_start, const int32_t* doy_rows, int max_window_rows,
                                      float q32, double q, int wd, float* thr_doy_major, int32_t* overflow) {
    if (!ctx) return -1;
    if (!anom || !doy_start || !doy_rows || !thr_doy_major || !overflow || T_out <= 0 || C <= 0 || max_window_rows <= 0)
        return fail(ctx, -1, "marex_hobday_exact_f32: null pointer or empty shape");
    if (wd < 1 || wd > 365 || (wd & 1) == 0) return fail(ctx, -1, "marex_hobday_exact_f32: window_days_hobday must be odd and in 1..365");
    if (!(q >= 0.0 && q <= 1.0)) return fail(ctx, -1, "marex_hobday_exact_f32: q must be in [0, 1]");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const int upper = q >= 0.5;
    const double tail = upper ? (1.0 - q) : q;
    int K = (int)ceil(tail * max_window_rows) + 4;
    if (K > max_window_rows) K = max_window_rows;
    int BT = 256;
    while (BT > 64 && (size_t)K * BT * 4 > 64 * 1024) BT >>= 1;
    const size_t lds = (size_t)K * BT * 4;
    if (lds > 64 * 1024)
        return fail(ctx, -4, "marex_hobday_exact_f32: window of %d samples at q=%.3f needs %zu bytes of LDS per workgroup", max_window_rows, q, lds);
    if (lds > 48 * 1024)
        HIP_TRY(ctx, hipFuncSetAttribute((const void*)k_hobday_exact, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    dim3 grid((unsigned)((C + BT - 1) / BT), NDOY);
    {
        LaunchTimer lt(ctx, MAREX_K_EXACT);
        hipLaunchKernelGGL(k_hobday_exact, grid, dim3(BT), lds, ctx->stream, anom, (long)C, doy_start, doy_rows, wd, q32,
                           upper, K, thr_doy_major, overflow);
    }
    HIP_TRY(ctx, hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------------------------------------
// K_G: global (constant in time) thresholds (detect.py:2737-2923) and the matching mask.
//  exact  : per cell, the two order statistics of ALL finite anomalies by an 8-bit radix select on the
//           order-preserving uint32 key (per-lane 256-bin LDS histogram, 4 passes per rank), float64 lerp
//           (DataArray.quantile -> np.nanquantile with a float64 q array).
//  approx : per cell 1-D histogram on float64 edges (per-lane nb-bin LDS column) and the CDF rule of
//           _compute_histogram_quantile_1d.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float key_to_float(unsigned k) {
    return __uint_as_float((k & 0x80000000u) ? (k & 0x7FFFFFFFu) : ~k);
}

#define GX_LANES 64
__global__ void __launch_bounds__(GX_LANES)
k_global_exact(const float* __restrict__ anom, long T, long C, double q, double* __restrict__ thr) {
    extern __shared__ unsigned rhist[];  // [256][GX_LANES]
    const long c = (long)blockIdx.x * GX_LANES + threadIdx.x;
    if (c >= C) return;
    unsigned* h = rhist + threadIdx.x;
    long m = 0;
    for (long t = 0; t < T; ++t) {
        const float v = anom[(size_t)t * C + c];
        m += (v == v);
    }
    if (m == 0) {
        thr[c] = __longlong_as_double(0x7FF8000000000000ll);
        return;
    }
    const double virt = (double)(m - 1) * q;
    long lo = (long)floor(virt);
    double g = virt - (double)lo;
    long hi = lo + 1;
    if (lo >= m - 1) {
        lo = m - 1;
        hi = m - 1;
    }
    float ab[2];
    for (int which = 0; which < 2; ++which) {
        long rank = which == 0 ? lo : hi;  // 0-based ascending rank
        if (which == 1 && hi == lo) {
            ab[1] = ab[0];
            break;
        }
        unsigned prefix = 0, pmask = 0;
        for (int pass = 0; pass < 4; ++pass) {
            const int sh = 24 - 8 * pass;
            for (int b = 0; b < 256; ++b) h[(size_t)b * GX_LANES] = 0u;
            for (long t = 0; t < T; ++t) {
                const float v = anom[(size_t)t * C + c];
                if (!(v == v)) continue;
                const unsigned k = ordered_key(v);
                if ((k & pmask) == prefix) h[(size_t)((k >> sh) & 255u) * GX_LANES] += 1u;
            }
            int b = 0;
            for (; b < 255; ++b) {
                const unsigned n = h[(size_t)b * GX_LANES];
                if ((unsigned long long)rank < n) break;
                rank -= n;
            }
            prefix |= (unsigned)b << sh;
            pmask |= 255u << sh;
        }
        ab[which] = key_to_float(prefix);
    }
    const double a = (double)ab[0], b = (double)ab[1];
    const double dba = (double)(ab[1] - ab[0]);  // float32 subtraction as in NumPy's _lerp
    double r = a + dba * g;
    if (g >= 0.5) r = b - dba * (1.0 - g);
    thr[c] = r;
}

#define GA_LANES 32
__global__ void __launch_bounds__(GA_LANES)
k_global_approx(const float* __restrict__ anom, long T, long C, const double* __restrict__ edges,
                const double* __restrict__ centres, int nb, double q, double lower_bound, double upper_bound,
                double* __restrict__ thr, marex_thr_stats* __restrict__ stats, double* __restrict__ minmax) {
    extern __shared__ unsigned ghist[];  // [nb][GA_LANES]
    const long c = (long)blockIdx.x * GA_LANES + threadIdx.x;
    if (c >= C) return;
    unsigned* h = ghist + threadIdx.x;
    for (int b = 0; b < nb; ++b) h[(size_t)b * GA_LANES] = 0u;
    const double e1 = edges[1], elast = edges[nb];
    const double inv_width = (double)(nb - 1) / (elast - e1);
    bool any_nan = false;
    for (long t = 0; t < T; ++t) {
        const float vf = anom[(size_t)t * C + c];
        if (!(vf == vf)) {
            any_nan = true;
            continue;
        }
        const double v = (double)vf;
        int k;
        if (v > elast) continue;            // beyond the last edge: not counted
        if (v == elast) k = nb - 1;         // right edge belongs to the last bin (np.histogram rule)
        else if (v < e1) k = 0;
        else {
            k = 1 + (int)((v - e1) * inv_width);
            k = k < 1 ? 1 : (k > nb - 1 ? nb - 1 : k);
            while (k > 1 && v < edges[k]) --k;
            while (k < nb - 1 && v >= edges[k + 1]) ++k;
        }
        h[(size_t)k * GA_LANES] += 1u;
    }
    double hsum = 0.0;
    for (int b = 0; b < nb; ++b) hsum += (double)h[(size_t)b * GA_LANES];
    hsum += 1e-10;
    const double eps = 1e-10;
    // cdf = cumsum(hist / hsum); first bin with cdf >= q - eps  (argmax of an all-False mask is 0)
    int iu = 0;
    {
        double cdf = 0.0;
        bool found = false;
        for (int b = 0; b < nb; ++b) {
            cdf += (double)h[(size_t)b * GA_LANES] / hsum;
            if (!found && cdf >= q - eps) {
                iu = b;
                found = true;
            }
        }
    }
    const int ib = (iu - 1 > 0) ? iu - 1 : 0;
    double cdf_t = 0.0;
    {
        double cdf = 0.0;
        for (int b = 0; b <= ib; ++b) cdf += (double)h[(size_t)b * GA_LANES] / hsum;
        cdf_t = cdf;
    }
    int il = 0;
    {
        double cdf = 0.0;
        bool found = false;
        for (int b = 0; b < nb; ++b) {
            cdf += (double)h[(size_t)b * GA_LANES] / hsum;
            if (!found && cdf > cdf_t) {
                il = b;
                found = true;
            }
        }
    }
    if (il > nb - 2) il = nb - 2;
    if (iu < 1) iu = 1;
    if (iu > nb - 1) iu = nb - 1;
    double cl = 0.0, cu = 0.0;
    {
        double cdf = 0.0;
        for (int b = 0; b < nb; ++b) {
            cdf += (double)h[(size_t)b * GA_LANES] / hsum;
            if (b == il) cl = cdf;
            if (b == iu) cu = cdf;
        }
    }
    const double bl = centres[il], bu = centres[iu];
    const double denom = cu - cl;
    const bool exact = fabs(cl - q) < eps, zero = fabs(denom) <= eps;
    const double frac = (q - cl) / (fabs(denom) > eps ? denom : 1.0);
    double r = bl + frac * (bu - bl);
    if (exact) r = bl;
    if (zero && !exact) r = (bl + bu) / 2;
    if (any_nan) r = __longlong_as_double(0x7FF8000000000000ll);
    if (r == r) {
        // min / max of the un-clamped thresholds for the warning text: float64 compare-and-swap loops
        unsigned long long* pmin = (unsigned long long*)&minmax[0];
        unsigned long long* pmax = (unsigned long long*)&minmax[1];
        unsigned long long old = *pmin;
        while (r < __longlong_as_double((long long)old)) {
            const unsigned long long seen = atomicCAS(pmin, old, (unsigned long long)__double_as_longlong(r));
            if (seen == old) break;
            old = seen;
        }
        old = *pmax;
        while (r > __longlong_as_double((long long)old)) {
            const unsigned long long seen = atomicCAS(pmax, old, (unsigned long long)__double_as_longlong(r));
            if (seen == old) break;
            old = seen;
        }
        if (r > upper_bound) atomicAdd(&stats->n_too_high, 1u);
        if (r < lower_bound) {
            atomicAdd(&stats->n_too_low, 1u);
            r = lower_bound;
        }
    }
    thr[c] = r;
}

// ------------------------------------------------------------------------------------------------
// Global thresholds, second generation (series of at most 65 535 steps): one wave = 64 cells, per-lane histograms
// with uint16 counters packed two per dword and stored lane-interleaved ([dword][lane]: every LDS access of a wave is
// conflict-free whatever bins the lanes hit), 16 coalesced row loads in flight.  Same arithmetic as k_global_approx /
// k_global_exact, which remain the fallback for longer series.
// ------------------------------------------------------------------------------------------------
#define G2_LANES 64
#define G2_BATCH 16

__global__ void __launch_bounds__(G2_LANES)
k_global_approx16(const float* __restrict__ anom, long T, long C, const double* __restrict__ edges,
                  const double* __restrict__ centres, int nb, double q, double lower_bound, double upper_bound,
                  double* __restrict__ thr, marex_thr_stats* __restrict__ stats, double* __restrict__ minmax) {
    extern __shared__ unsigned g2[];
    const int nbw = (nb + 1) >> 1;
    unsigned* hist = g2;                                               // [nbw][64]
    double* led = reinterpret_cast<double*>(g2 + (size_t)nbw * G2_LANES);  // [nb + 1] edges
    const int lane = threadIdx.x;
    for (int i = lane; i <= nb; i += G2_LANES) led[i] = edges[i];
    for (int d = 0; d < nbw; ++d) hist[d * G2_LANES + lane] = 0u;
    __syncthreads();
    const long c = (long)blockIdx.x * G2_LANES + lane;
    if (c >= C) return;  // single wave, no barrier below
    auto H = [&](int b) { return (double)((hist[(b >> 1) * G2_LANES + lane] >> ((b & 1) * 16)) & 0xFFFFu); };
    const double e1 = led[1], elast = led[nb];
    const double inv_width = (double)(nb - 1) / (elast - e1);
    bool any_nan = false;
    auto count = [&](float vf) {
        if (!(vf == vf)) {
            any_nan = true;
            return;
        }
        const double v = (double)vf;
        int k;
        if (v > elast) return;              // beyond the last edge: not counted
        if (v == elast) k = nb - 1;         // right edge belongs to the last bin (np.histogram rule)
        else if (v < e1) k = 0;
        else {
            k = 1 + (int)((v - e1) * inv_width);
            k = k < 1 ? 1 : (k > nb - 1 ? nb - 1 : k);
            while (k > 1 && v < led[k]) --k;
            while (k < nb - 1 && v >= led[k + 1]) ++k;
        }
        hist[(k >> 1) * G2_LANES + lane] += 1u << ((k & 1) * 16);
    };
    long t = 0;
    for (; t + G2_BATCH <= T; t += G2_BATCH) {
        float v[G2_BATCH];
#pragma unroll
        for (int u = 0; u < G2_BATCH; ++u) v[u] = anom[(size_t)(t + u) * C + c];
#pragma unroll
        for (int u = 0; u < G2_BATCH; ++u) count(v[u]);
    }
    for (; t < T; ++t) count(anom[(size_t)t * C + c]);

    double hsum = 0.0;
    for (int b = 0; b < nb; ++b) hsum += H(b);
    hsum += 1e-10;
    const double eps = 1e-10;
    int iu = 0;
    {
        double cdf = 0.0;
        bool found = false;
        for (int b = 0; b < nb; ++b) {
            cdf += H(b) / hsum;
            if (!found && cdf >= q - eps) {
                iu = b;
                found = true;
            }
        }
    }
    const int ib = (iu - 1 > 0) ? iu - 1 : 0;
    double cdf_t = 0.0;
    {
        double cdf = 0.0;
        for (int b = 0; b <= ib; ++b) cdf += H(b) / hsum;
        cdf_t = cdf;
    }
    int il = 0;
    {
        double cdf = 0.0;
        bool found = false;
        for (int b = 0; b < nb; ++b) {
            cdf += H(b) / hsum;
            if (!found && cdf > cdf_t) {
                il = b;
                found = true;
            }
        }
    }
    if (il > nb - 2) il = nb - 2;
    if (iu < 1) iu = 1;
    if (iu > nb - 1) iu = nb - 1;
    double cl = 0.0, cu = 0.0;
    {
        double cdf = 0.0;
        for (int b = 0; b < nb; ++b) {
            cdf += H(b) / hsum;
            if (b == il) cl = cdf;
            if (b == iu) cu = cdf;
        }
    }
    const double bl = centres[il], bu = centres[iu];
    const double denom = cu - cl;
    const bool exact = fabs(cl - q) < eps, zero = fabs(denom) <= eps;
    const double frac = (q - cl) / (fabs(denom) > eps ? denom : 1.0);
    double r = bl + frac * (bu - bl);
    if (exact) r = bl;
    if (zero && !exact) r = (bl + bu) / 2;
    if (any_nan) r = __longlong_as_double(0x7FF8000000000000ll);
    if (r == r) {
        unsigned long long* pmin = (unsigned long long*)&minmax[0];
        unsigned long long* pmax = (unsigned long long*)&minmax[1];
        unsigned long long old = *pmin;
        while (r < __longlong_as_double((long long)old)) {
            const unsigned long long seen = atomicCAS(pmin, old, (unsigned long long)__double_as_longlong(r));
            if (seen == old) break;
            old = seen;
        }
        old = *pmax;
        while (r > __longlong_as_double((long long)old)) {
            const unsigned long long seen = atomicCAS(pmax, old, (unsigned long long)__double_as_longlong(r));
            if (seen == old) break;
            old = seen;
        }
        if (r > upper_bound) atomicAdd(&stats->n_too_high, 1u);
        if (r < lower_bound) {
            atomicAdd(&stats->n_too_low, 1u);
            r = lower_bound;
        }
    }
    thr[c] = r;
}

// np.nanquantile(x, q), "linear": values at the ascending ranks lo = floor((m-1) q) and lo + 1 among the m non-NaN
// samples.  Four 8-bit radix passes over an order-preserving key find rank lo (the first pass also counts m); the
// descent knows how many samples are <= that value, so rank lo + 1 is either the same value (ties) or the smallest
// larger sample, found by one more pass -- 5 passes over the series instead of 9.
__global__ void __launch_bounds__(G2_LANES)
k_global_exact16(const float* __restrict__ anom, long T, long C, double q, double* __restrict__ thr) {
    extern __shared__ unsigned g2[];  // [128][64]: 256 uint16 counters per lane
    const int lane = threadIdx.x;
    const long c = (long)blockIdx.x * G2_LANES + lane;
    if (c >= C) return;
    auto H = [&](int b) { return (hist_get(g2, b, lane)); };
    unsigned prefix = 0, pmask = 0;
    long m = 0, rank = 0, lo = 0, hi = 0, below = 0;  // below = samples smaller than the current prefix bucket
    double g = 0.0;
    unsigned n_final = 0;
    for (int pass = 0; pass < 4; ++pass) {
        const int sh = 24 - 8 * pass;
        for (int d = 0; d < 128; ++d) g2[d * G2_LANES + lane] = 0u;
        auto count = [&](float v) {
            if (!(v == v)) return;
            const unsigned k = ordered_key(v);
            if ((k & pmask) == prefix) {
                const unsigned b = (k >> sh) & 255u;
                g2[(b >> 1) * G2_LANES + lane] += 1u << ((b & 1u) * 16);
            }
        };
        long t = 0;
        for (; t + G2_BATCH <= T; t += G2_BATCH) {
            float v[G2_BATCH];
#pragma unroll
            for (int u = 0; u < G2_BATCH; ++u) v[u] = anom[(size_t)(t + u) * C + c];
#pragma unroll
            for (int u = 0; u < G2_BATCH; ++u) count(v[u]);
        }
        for (; t < T; ++t) count(anom[(size_t)t * C + c]);
        if (pass == 0) {
            for (int b = 0; b < 256; ++b) m += H(b);
            if (m == 0) {
                thr[c] = __longlong_as_double(0x7FF8000000000000ll);
                return;
            }
            const double virt = (double)(m - 1) * q;
            lo = (long)floor(virt);
            g = virt - (double)lo;
            hi = lo + 1;
            if (lo >= m - 1) {
                lo = m - 1;
                hi = m - 1;
            }
            rank = lo;
        }
        int b = 0;
        for (; b < 255; ++b) {
            const unsigned n = H(b);
            if ((unsigned long long)rank < n) break;
            rank -= n;
            below += n;
        }
        n_final = H(b);
        prefix |= (unsigned)b << sh;
        pmask |= 255u << sh;
    }
    const float a = key_to_float(prefix);
    float bv = a;
    if (hi != lo && hi >= below + (long)n_final) {  // rank lo + 1 is not another copy of a: smallest larger sample
        unsigned best = 0xFFFFFFFFu;
        auto look = [&](float v) {
            if (!(v == v)) return;
            const unsigned k = ordered_key(v);
            if (k > prefix && k < best) best = k;
        };
        long t = 0;
        for (; t + G2_BATCH <= T; t += G2_BATCH) {
            float v[G2_BATCH];
#pragma unroll
            for (int u = 0; u < G2_BATCH; ++u) v[u] = anom[(size_t)(t + u) * C + c];
#pragma unroll
            for (int u = 0; u < G2_BATCH; ++u) look(v[u]);
        }
        for (; t < T; ++t) look(anom[(size_t)t * C + c]);
        bv = key_to_float(best);
    }
    const double ad = (double)a, bd = (double)bv;
    const double dba = (double)(bv - a);  // float32 subtraction as in NumPy's _lerp
    double r = ad + dba * g;
    if (g >= 0.5) r = bd - dba * (1.0 - g);
    thr[c] = r;
}

extern "C" int marex_global_threshold_f32(marex_ctx* ctx, const float* anom, int64_t T_out, int64_t C, double q,
                                          int exact, const double* edges, const double* centres, int nb,
                                          double lower_bound, double upper_bound, double* thr,
                                          marex_thr_stats* stats, double* minmax) {
    if (!ctx) return -1;
    if (!anom || !thr || T_out <= 0 || C <= 0) return fail(ctx, -1, "marex_global_threshold_f32: null pointer or empty shape");
    if (!(q >= 0.0 && q <= 1.0)) return fail(ctx, -1, "marex_global_threshold_f32: q must be in [0, 1]");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    LaunchTimer lt(ctx, MAREX_K_GLOBAL);
    const bool small_counts = T_out <= 65535 && env_int("MAREX_GLOBAL_V1", 0) == 0;  // uint16 counters suffice
    if (exact && small_counts) {
        const size_t lds = 128 * G2_LANES * 4;  // 32 KiB
        hipLaunchKernelGGL(k_global_exact16, dim3((unsigned)((C + G2_LANES - 1) / G2_LANES)), dim3(G2_LANES), lds,
                           ctx->stream, anom, (long)T_out, (long)C, q, thr);
    } else if (exact) {
        const size_t lds = 256 * GX_LANES * 4;  // 64 KiB
        HIP_TRY(ctx, hipFuncSetAttribute((const void*)k_global_exact, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(k_global_exact, dim3((unsigned)((C + GX_LANES - 1) / GX_LANES)), dim3(GX_LANES), lds,
                           ctx->stream, anom, (long)T_out, (long)C, q, thr);
    } else {
        if (!edges || !centres || !stats || !minmax || nb < 4 || nb > 600)
            return fail(ctx, -1, "marex_global_threshold_f32: approximate method needs edges, centres, stats, minmax and 4 <= nb <= 600");
        if (small_counts) {
            const size_t lds2 = (size_t)((nb + 1) / 2) * G2_LANES * 4 + (size_t)(nb + 1) * 8;  // <= 82 KiB for nb <= 600
            if (lds2 > 48 * 1024)
                HIP_TRY(ctx, hipFuncSetAttribute((const void*)k_global_approx16, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2));
            hipLaunchKernelGGL(k_global_approx16, dim3((unsigned)((C + G2_LANES - 1) / G2_LANES)), dim3(G2_LANES), lds2,
                               ctx->stream, anom, (long)T_out, (long)C, edges, centres, nb, q, lower_bound, upper_bound, thr,
                               stats, minmax);
            HIP_TRY(ctx, hipGetLastError());
            return 0;
        }
        const size_t lds = (size_t)nb * GA_LANES * 4;  // <= 75 KiB for nb <= 600
        if (lds > 48 * 1024)
            HIP_TRY(ctx, hipFuncSetAttribute((const void*)k_global_approx, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(k_global_approx, dim3((unsigned)((C + GA_LANES - 1) / GA_LANES)), dim3(GA_LANES), lds, ctx->stream, anom, (long)T_out,
                           (long)C, edges, centres, nb, q, lower_bound, upper_bound, thr, stats, minmax);
    }
    HIP_TRY(ctx, hipGetLastError());
    return 0;
}

// extreme[t, c] = (double)anom[t, c] >= thr[c]   (detect.py:2915)
__global__ void __launch_bounds__(256)
k_mask_ge_const(const float* __restrict__ anom, const double* __restrict__ thr, long T, long C, int rows_per_block,
                unsigned char* __restrict__ out, unsigned long long* __restrict__ n_true) {
    const long c = (long)blockIdx.x * 256 + threadIdx.x;
    unsigned cnt = 0;
    if (c < C) {
        const double th = thr[c];
        const long t0 = (long)blockIdx.y * rows_per_block;
        const long t1 = t0 + rows_per_block < T ? t0 + rows_per_block : T;
        for (long t = t0; t < t1; ++t) {
            const unsigned char m = (double)anom[(size_t)t * C + c] >= th;
            cnt += m;
            out[(size_t)t * C + c] = m;
        }
    }
    if (n_true) {
        for (int s = 32; s > 0; s >>= 1) cnt += __shfl_down(cnt, s, 64);
        if ((threadIdx.x & 63) == 0 && cnt) atomicAdd(n_true, (unsigned long long)cnt);
    }
}

// four cells per lane: 16-byte anomaly loads, 4-byte mask stores, four rows in flight
__global__ void __launch_bounds__(256)
k_mask_ge_const4(const float* __restrict__ anom, const double* __restrict__ thr, long T, long C, int rows_per_block,
                 unsigned char* __restrict__ out, unsigned long long* __restrict__ n_true) {
    const long c = ((long)blockIdx.x * 256 + threadIdx.x) * 4;
    unsigned cnt = 0;
    if (c < C) {
        const double t0d = thr[c], t1d = thr[c + 1], t2d = thr[c + 2], t3d = thr[c + 3];
        const long t0 = (long)blockIdx.y * rows_per_block;
        const long t1 = t0 + rows_per_block < T ? t0 + rows_per_block : T;
        auto one = [&](long t, float4 a) {
            uchar4 m;
            m.x = (double)a.x >= t0d;
            m.y = (double)a.y >= t1d;
            m.z = (double)a.z >= t2d;
            m.w = (double)a.w >= t3d;
            cnt += m.x + m.y + m.z + m.w;
            *reinterpret_cast<uchar4*>(out + (size_t)t * C + c) = m;
        };
        long t = t0;
        for (; t + 4 <= t1; t += 4) {
            float4 a[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) a[u] = *reinterpret_cast<const float4*>(anom + (size_t)(t + u) * C + c);
#pragma unroll
            for (int u = 0; u < 4; ++u) one(t + u, a[u]);
        }
        for (; t < t1; ++t) one(t, *reinterpret_cast<const float4*>(anom + (size_t)t * C + c));
    }
    if (n_true) {
        for (int s = 32; s > 0; s >>= 1) cnt += __shfl_down(cnt, s, 64);
        if ((threadIdx.x & 63) == 0 && cnt) atomicAdd(n_true, (unsigned long long)cnt);
    }
}

extern "C" int marex_mask_ge_const_f32(marex_ctx* ctx, const float* anom, const double* thr, int64_t T_out, int64_t C,
                                       uint8_t* extreme, unsigned long long* n_true) {
    if (!ctx) return -1;
    if (!anom || !thr || !extreme || T_out <= 0 || C <= 0) return fail(ctx, -1, "marex_mask_ge_const_f32: bad argument");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const int rows = 64;
    {
        LaunchTimer lt(ctx, MAREX_K_MASK);
        if ((C & 3) == 0) {
            dim3 grid((unsigned)((C / 4 + 255) / 256), (unsigned)((T_out + rows - 1) / rows));
            hipLaunchKernelGGL(k_mask_ge_const4, grid, dim3(256), 0, ctx->stream, anom, thr, (long)T_out, (long)C, rows, extreme, n_true);
        } else {
            dim3 grid((unsigned)((C + 255) / 256), (unsigned)((T_out + rows - 1) / rows));
            hipLaunchKernelGGL(k_mask_ge_const, grid, dim3(256), 0, ctx->stream, anom, thr, (long)T_out, (long)C, rows, extreme, n_true);
        }
    }
    HIP_TRY(ctx, hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------------------------------------
// transpose (thresholds [366, C] -> [C, 366])
// ------------------------------------------------------------------------------------------------
// std_normalise (detect.py:2257-2278): day-of-year standard deviation, wrapped rolling RMS, division
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
k_doy_std(const float* __restrict__ anom, const int* __restrict__ doy_start, const int* __restrict__ doy_rows, long C,
          float* __restrict__ std_day) {
    const long c = (long)blockIdx.x * 256 + threadIdx.x;
    const int d = blockIdx.y;
    if (c >= C) return;
    const int r0 = doy_start[d], r1 = doy_start[d + 1];
    float res = nan_f();
    if (r1 > r0) {
        double sum = 0.0;
        for (int r = r0; r < r1; ++r) sum += (double)anom[(size_t)doy_rows[r] * C + c];
        const double mean = sum / (double)(r1 - r0);
        double ss = 0.0;
        for (int r = r0; r < r1; ++r) {  // second pass over the same few rows (L2 resident)
            const double dv = (double)anom[(size_t)doy_rows[r] * C + c] - mean;
            ss += dv * dv;
        }
        res = (float)sqrt(ss / (double)(r1 - r0));
    }
    std_day[(size_t)d * C + c] = res;
}

__global__ void __launch_bounds__(256)
k_std_rolling(const float* __restrict__ std_day, long C, int window, float* __restrict__ std_roll) {
    const long c = (long)blockIdx.x * 256 + threadIdx.x;
    const int d = blockIdx.y;
    if (c >= C) return;
    const int lo = window / 2;
    double acc = 0.0;
    for (int k = 0; k < window; ++k) {
        int dd = (d - lo + k) % NDOY;
        if (dd < 0) dd += NDOY;
        const float sd = std_day[(size_t)dd * C + c];
        const float sq = sd * sd;
        acc += (double)sq;
    }
    const float m = (float)(acc / (double)window);
    std_roll[(size_t)d * C + c] = sqrtf(m);
}

__global__ void __launch_bounds__(256)
k_div_doy(const float* __restrict__ anom, const float* __restrict__ std_roll, const int* __restrict__ doy_start,
          const int* __restrict__ doy_rows, long C, float* __restrict__ out) {
    const int dA = (int)blockIdx.y * NDOY / MASK_DOY_CHUNKS, dB = ((int)blockIdx.y + 1) * NDOY / MASK_DOY_CHUNKS;
    const long c = (long)blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    for (int d = dA; d < dB; ++d) {
        const float sd = std_roll[(size_t)d * C + c];
        const float safe = sd > 1e-10f ? sd : nan_f();
        const int r0 = doy_start[d], r1 = doy_start[d + 1];
        for (int r = r0; r < r1; ++r) {
            const size_t off = (size_t)doy_rows[r] * C + c;
            out[off] = anom[off] / safe;
        }
    }
}

extern "C" int marex_std_rolling_doy_f32(marex_ctx* ctx, const float* anom, int64_t T, int64_t C,
                                         const int32_t* doy_start, const int32_t* doy_rows, int window,
                                         float* std_day, float* std_roll) {
    if (!ctx) return -1;
    if (!anom || !doy_start || !doy_rows || !std_day || !std_roll || T <= 0 || C <= 0)
        return fail(ctx, -1, "marex_std_rolling_doy_f32: null pointer or empty shape");
    if (window < 1 || window > NDOY) return fail(ctx, -1, "marex_std_rolling_doy_f32: window must be in 1..366");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    dim3 grid((unsigned)((C + 255) / 256), NDOY);
    {
        LaunchTimer lt(ctx, MAREX_K_STDNORM);
        hipLaunchKernelGGL(k_doy_std, grid, dim3(256), 0, ctx->stream, anom, doy_start, doy_rows, (long)C, std_day);
        hipLaunchKernelGGL(k_std_rolling, grid, dim3(256), 0, ctx->stream, std_day, (long)C, window, std_roll);
    }
    HIP_TRY(ctx, hipGetLastError());
    return 0;
}

extern "C" int marex_div_doy_f32(marex_ctx* ctx, const float* anom, const float* std_roll, const int32_t* doy_start,
                                 const int32_t* doy_rows, int64_t T, int64_t C, float* out) {
    if (!ctx) return -1;
    if (!anom || !std_roll || !doy_start || !doy_rows || !out || T <= 0 || C <= 0)
        return fail(ctx, -1, "marex_div_doy_f32: null pointer or empty shape");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    dim3 grid((unsigned)((C + 255) / 256), MASK_DOY_CHUNKS);
    {
        LaunchTimer lt(ctx, MAREX_K_STDNORM);
        hipLaunchKernelGGL(k_div_doy, grid, dim3(256), 0, ctx->stream, anom, std_roll, doy_start, doy_rows, (long)C, out);
    }
    HIP_TRY(ctx, hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_transpose(const float* __restrict__ in, long rows, long cols,
                                                   float* __restrict__ out) {
    __shared__ float tile[32][33];
    const long c0 = (long)blockIdx.x * 32, r0 = (long)blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
    for (int k = ty; k < 32; k += 8) {
        const long r = r0 + k, c = c0 + tx;
        if (r < rows && c < cols) tile[k][tx] = in[(size_t)r * cols + c];
    }
    __syncthreads();
    for (int k = ty; k < 32; k += 8) {
        const long c = c0 + k, r = r0 + tx;
        if (r < rows && c < cols) out[(size_t)c * rows + r] = tile[tx][k];
    }
}

extern "C" int marex_transpose_f32(marex_ctx* ctx, const float* in, int64_t rows, int64_t cols, float* out) {
    if (!ctx) return -1;
    if (!in || !out || rows <= 0 || cols <= 0) return fail(ctx, -1, "marex_transpose_f32: bad argument");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    dim3 grid((unsigned)((cols + 31) / 32), (unsigned)((rows + 31) / 32));
    {
        LaunchTimer lt(ctx, MAREX_K_TRANSPOSE);
        hipLaunchKernelGGL(k_transpose, grid, dim3(256), 0, ctx->stream, in, (long)rows, (long)cols, out);
    }
    HIP_TRY(ctx, hipGetLastError());
    return 0;
}
