// marex_anomalies.hip -- fixed baseline, digitize, detrend, std_normalise
#include "marex_common.hip.h"
#include "marex_tails.hip.h"

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
                 int* __restrict__ invalid_count, const float* __restrict__ sub) {
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
    // sub: a per-cell value taken off every sample on load (the residual mean of a detrend whose last pass -- subtracting
    // it from the whole field -- is folded in here: the same single float32 subtraction, one round trip through HBM less)
    const bool has_sub = sub != nullptr;
    const float sc = has_sub ? sub[c] : 0.f;
    float acc = 0.f;
    int n = 0, n_invalid = 0;
    for (int r = r0; r < r1; ++r) {
        const int t = doy_rows[r];
        float v = x[(size_t)t * C + c];
        if (has_sub) v -= sc;
        if (!finite_f(v)) ++n_invalid;
        if ((!use_row || use_row[t]) && v == v) {
            acc += v;
            ++n;
        }
    }
    const float clim = acc / (float)n;  // n == 0 -> NaN
    for (int r = r0; r < r1; ++r) {
        const int t = doy_rows[r];
        float v = x[(size_t)t * C + c];
        if (has_sub) v -= sc;
        const float a = v - clim;
        out[(size_t)t * C + c] = a;
        if (do_bins) bins[bins_index(r, c, T)] = (unsigned short)digitize_bin(a, e, nb, inv_width);
    }
    if (invalid_count && n_invalid) atomicAdd(&invalid_count[c], n_invalid);
}

// The same without a bin matrix, for dayofyear buckets of at most NMAX rows: a lane keeps its bucket in REGISTERS between the
// sum and the subtraction, so the field is read once (k_fixed_baseline's second read of a 100-year bucket does not come from
// the L2: 205 MB of buckets are in flight at full occupancy -- measured 39 GB of reads for a 19.7 GB band) and up to NMAX
// loads per lane are in flight (two waves per SIMD are plenty for a stream like that).
// NC > 0: x is the RAW field and the sample is its detrend residual, x - fl32(sum_k model_t[t][k] * coef[k][c]) (the arithmetic of
// k_detrend_resid, recomputed here so that the residual field is never written or read: detrend_fixed_baseline in 3 reads and
// 1 write of the field).
// TAILS: the kernel also leaves the sorted key lists of its own output (marex_tails.hip.h, lists of 32 rows: what
// k_tail_extract would make of `out`), so that the threshold and mask kernels need no extraction pass -- a read of the whole
// anomaly field and 13 ms per 100-yr band of the detrend_fixed_baseline path (round 3).  A lane holds ONE bucket, so the packed
// 16-bit sorting networks take two LISTS of that bucket per register (rows 64 P + u in the low halves, rows 64 P + 32 + u in the
// high ones) instead of two dayofyears: 2 x 206 compare-exchanges for a 100-row bucket, hidden behind the kernel's memory time.
template <int NMAX, int NC = 0, bool TAILS = false>
__global__ void __launch_bounds__(256)
k_fixed_baseline_reg(const float* __restrict__ x, long C, const int* __restrict__ doy_start, const int* __restrict__ doy_rows,
                     const unsigned char* __restrict__ use_row, float* __restrict__ out, unsigned char* __restrict__ mask,
                     int* __restrict__ invalid_count, const float* __restrict__ sub, const double* __restrict__ model_t = nullptr,
                     const double* __restrict__ coef = nullptr, const float* __restrict__ edges = nullptr, int nb = 0,
                     uint4* __restrict__ lists = nullptr, unsigned* __restrict__ aux = nullptr, int NPER_ALL = 0) {
    extern __shared__ float e_lds[];  // TAILS: [nb + 1] edge table
    bool lean = false;
    float e_first = 0.f, e_delta = 0.f, e_last = 0.f, inv_width = 0.f, c0 = 0.f, nbm1f = 0.f;
    if (TAILS) {
        for (int i = threadIdx.x; i <= nb; i += 256) e_lds[i] = edges[i];
        __syncthreads();
        e_first = e_lds[1], e_delta = e_lds[2] - e_lds[1], e_last = e_lds[nb];
        inv_width = (float)(nb - 1) / (e_last - e_first);
        lean = edges_are_arange(e_lds, nb);  // the anomaly kernels' digitize (marex_tails.hip: k_tail_extract)
        const double m = fabs((double)e_first) > fabs((double)e_last) ? fabs((double)e_first) : fabs((double)e_last);
        lean = lean && e_delta > 0.f && ((double)nb + 2.0 * m / (double)e_delta) * (1.0 / 1048576.0) < 1.0 / 256.0;
        c0 = (1.0f - e_first * inv_width) - 0.0078125f;
        nbm1f = (float)(nb - 1);
    }
    const int d = blockIdx.y;
    const long c_raw = (long)blockIdx.x * 256 + threadIdx.x;
    const bool active = c_raw < C;
    const long c = active ? c_raw : C - 1;
    if (d == 0 && mask && active) mask[c] = finite_f(x[c]) ? 1 : 0;
    const int r0 = doy_start[d], nrow = doy_start[d + 1] - r0;  // uniform, <= NMAX
    const bool has_sub = sub != nullptr;
    const float sc = has_sub ? sub[c] : 0.f;
    double cf[NC > 0 ? NC : 1];
#pragma unroll
    for (int k = 0; k < NC; ++k) cf[k] = coef[(size_t)k * C + c];
    float v[NMAX];
#pragma unroll
    for (int b = 0; b < NMAX; b += 16) {
        if (b < nrow) {
#pragma unroll
            for (int u = 0; u < 16; ++u)
                if (b + u < nrow) v[b + u] = x[(size_t)doy_rows[r0 + b + u] * C + c];
        }
    }
    if (NC > 0) {  // a second walk, so that the loads above are all in flight before the first residual is formed
#pragma unroll
        for (int b = 0; b < NMAX; b += 16) {
            if (b < nrow) {
#pragma unroll
                for (int u = 0; u < 16; ++u)
                    if (b + u < nrow) {
                        // model rows in BUCKET order (model_t here = the caller's table permuted by doy_rows): the rows of a
                        // batch are contiguous, one wide scalar load instead of a dependent load per row
                        const double* mt = model_t + (size_t)(r0 + b + u) * NC;
                        double trend = 0.0;
#pragma unroll
                        for (int k = 0; k < NC; ++k) trend += mt[k] * cf[k];
                        v[b + u] = v[b + u] - (float)trend;
                    }
            }
        }
    }
    float acc = 0.f;
    int n = 0, n_invalid = 0;
#pragma unroll
    for (int b = 0; b < NMAX; b += 16) {
        if (b < nrow) {
#pragma unroll
            for (int u = 0; u < 16; ++u)
                if (b + u < nrow) {
                    float w = v[b + u];
                    if (has_sub) w -= sc;
                    v[b + u] = w;
                    if (!finite_f(w)) ++n_invalid;
                    if ((!use_row || use_row[doy_rows[r0 + b + u]]) && w == w) {  // ascending time: the oracle's order
                        acc += w;
                        ++n;
                    }
                }
        }
    }
    const float clim = acc / (float)n;  // n == 0 -> NaN
#pragma unroll
    for (int b = 0; b < NMAX; b += 16) {
        if (b < nrow) {
#pragma unroll
            for (int u = 0; u < 16; ++u)
                if (b + u < nrow) {
                    v[b + u] = v[b + u] - clim;  // from here on the registers hold the anomalies (TAILS reads them again)
                    if (active) out[(size_t)doy_rows[r0 + b + u] * C + c] = v[b + u];
                }
        }
    }
    if (invalid_count && n_invalid && active) atomicAdd(&invalid_count[c], n_invalid);
    if constexpr (TAILS) {
        constexpr int NPAIRS = (NMAX + 63) / 64;  // pairs of 32-row lists
        // NPER_ALL: lists per bucket of the buffer (ceil(max_bucket / 32)); lists beyond this bucket's rows are written empty
        unsigned cnt = 0, ovf = 0;
        auto key_of = [&](float a, int pos) -> unsigned {
            int k;
            if (lean) {
                const float f = __builtin_fmaf(a, inv_width, c0);
                const float t = __builtin_amdgcn_fmed3f(__builtin_floorf(f), 0.0f, nbm1f);
                const float ehi = e_first + t * e_delta;
                k = (int)t + (a >= ehi ? 1 : 0);
                k = (a == a) ? k : nb;
            } else {
                k = digitize_bin(a, e_lds, nb, inv_width);
            }
            const bool ok = k < nb;
            cnt += ok ? 1u : 0u;
            return ok ? tail_key(k, pos) : 0u;
        };
        // one pair of lists (rows 64 P .. 64 P + 63) -- as a macro per P: a loop over P with this body is not unrolled by hipcc,
        // and a runtime P indexes the register array v[] (which then lives in scratch memory: 528 bytes per lane)
#define MAREX_FB_PAIR(P)                                                                                          \
    if (2 * (P) < NPER_ALL) {                                                                                     \
            unsigned pk[32]; \
            float amax = -__builtin_inff(); \
_Pragma("unroll") \
            for (int u = 0; u < 32; ++u) { \
                const int ra = 64 * P + u, rb = 64 * P + 32 + u; \
                unsigned ka = 0u, kb = 0u; \
                if (ra < NMAX && ra < nrow) { \
                    ka = key_of(v[ra < NMAX ? ra : 0], ra); \
                    amax = fmaxf(amax, v[ra < NMAX ? ra : 0]); \
                } \
                if (rb < NMAX && rb < nrow) { \
                    kb = key_of(v[rb < NMAX ? rb : 0], rb); \
                    amax = fmaxf(amax, v[rb < NMAX ? rb : 0]); \
                } \
                pk[u] = ka | (kb << 16); \
            } \
            if (__builtin_amdgcn_ballot_w64(amax >= e_last) != 0) { \
_Pragma("unroll") \
                for (int r = 64 * P; r < 64 * P + 64; ++r) \
                    if (r < NMAX && r < nrow && v[r < NMAX ? r : 0] >= e_last) ovf = tail_ovf_add(ovf, (unsigned)r); \
            } \
            sort16_desc(reinterpret_cast<unsigned(&)[16]>(pk[0])); \
            sort16_desc(reinterpret_cast<unsigned(&)[16]>(pk[16])); \
_Pragma("unroll") \
            for (int i = 0; i < 8; ++i) { \
                const unsigned tmp = pk[16 + i]; \
                pk[16 + i] = pk[31 - i]; \
                pk[31 - i] = tmp; \
            } \
            bitonic_merge_desc<32>(pk); \
            if (active) { \
_Pragma("unroll") \
                for (int j = 0; j < 4; ++j) { \
                    uint4 w0, w1; \
                    unsigned* a = reinterpret_cast<unsigned*>(&w0); \
                    unsigned* b = reinterpret_cast<unsigned*>(&w1); \
_Pragma("unroll") \
                    for (int i = 0; i < 4; ++i) { \
                        const unsigned xk = pk[8 * j + 2 * i], yk = pk[8 * j + 2 * i + 1]; \
                        a[i] = (xk & 0xFFFFu) | (yk << 16); \
                        b[i] = (xk >> 16) | (yk & 0xFFFF0000u); \
                    } \
                    if (2 * P < NPER_ALL) lists[(((size_t)d * NPER_ALL + 2 * P) * 4 + j) * C + c] = w0; \
                    if (2 * P + 1 < NPER_ALL) lists[(((size_t)d * NPER_ALL + 2 * P + 1) * 4 + j) * C + c] = w1; \
                } \
            } \
    }
        MAREX_FB_PAIR(0)
        if constexpr (NPAIRS > 1) { MAREX_FB_PAIR(1) }
#undef MAREX_FB_PAIR
        if (active) aux[(size_t)d * C + c] = tail_aux_word(cnt, ovf);
    }
}

static int fixed_baseline_impl(marex_ctx* ctx, const float* x, int64_t T, int64_t C, const int32_t* doy_start,
                               const int32_t* doy_rows, const uint8_t* use_row, const float* edges, int nb, float* out,
                               uint16_t* bins, uint8_t* mask, int32_t* invalid_count, const float* sub, int max_bucket);

extern "C" int marex_fixed_baseline_f32(marex_ctx* ctx, const float* x, int64_t T, int64_t C,
                                        const int32_t* doy_start, const int32_t* doy_rows,
                                        const uint8_t* use_row, const float* edges, int nb, float* out,
                                        uint16_t* bins, uint8_t* mask, int32_t* invalid_count) {
    return fixed_baseline_impl(ctx, x, T, C, doy_start, doy_rows, use_row, edges, nb, out, bins, mask, invalid_count, nullptr, 0);
}

extern "C" int marex_fixed_baseline_sub_f32(marex_ctx* ctx, const float* x, const float* sub, int max_bucket, int64_t T, int64_t C,
                                            const int32_t* doy_start, const int32_t* doy_rows, const uint8_t* use_row,
                                            const float* edges, int nb, float* out, uint16_t* bins, uint8_t* mask,
                                            int32_t* invalid_count) {
    return fixed_baseline_impl(ctx, x, T, C, doy_start, doy_rows, use_row, edges, nb, out, bins, mask, invalid_count, sub, max_bucket);
}

static int fixed_baseline_impl(marex_ctx* ctx, const float* x, int64_t T, int64_t C, const int32_t* doy_start,
                               const int32_t* doy_rows, const uint8_t* use_row, const float* edges, int nb, float* out,
                               uint16_t* bins, uint8_t* mask, int32_t* invalid_count, const float* sub, int max_bucket) {
    if (!ctx) return -1;
    if (!x || !doy_start || !doy_rows || !out || T <= 0 || C <= 0)
        return fail(ctx, -1, "marex_fixed_baseline_f32: null pointer or empty shape");
    if (bins && (!edges || nb < 4 || nb > 36000)) return fail(ctx, -1, "marex_fixed_baseline_f32: binning needs edges and 4 <= nb <= 36000");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    dim3 grid((unsigned)((C + 255) / 256), NDOY);
    if (!bins && max_bucket >= 1 && max_bucket <= 128 && ctx_opt(ctx, "FIXED_REG", 1)) {  // buckets held in registers: one read
        LaunchTimer lt(ctx, MAREX_K_FIXED);
        if (max_bucket <= 48)
            hipLaunchKernelGGL(k_fixed_baseline_reg<48>, grid, dim3(256), 0, ctx->stream, x, (long)C, doy_start, doy_rows, use_row, out,
                               mask, invalid_count, sub);
        else
            hipLaunchKernelGGL(k_fixed_baseline_reg<128>, grid, dim3(256), 0, ctx->stream, x, (long)C, doy_start, doy_rows, use_row, out,
                               mask, invalid_count, sub);
        HIP_TRY(ctx, hipGetLastError());
        return 0;
    }
    const size_t lds = bins ? ((size_t)nb + 1) * sizeof(float) : 0;
    if (lds > 48 * 1024)
        HIP_TRY(ctx, hipFuncSetAttribute((const void*)k_fixed_baseline, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    {
        LaunchTimer lt(ctx, MAREX_K_FIXED);
        hipLaunchKernelGGL(k_fixed_baseline, grid, dim3(256), lds, ctx->stream, x, (long)T, (long)C, doy_start, doy_rows,
                           use_row, edges, nb, out, bins, mask, invalid_count, sub);
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
#ifndef DETREND_UNROLL
#define DETREND_UNROLL 16
#endif
#define DETREND_TBLOCK 1024  // timesteps per partial sum (arithmetic contract, oracle.DETREND_TBLOCK)

// Reductions over time are split into blocks of DETREND_TBLOCK timesteps so that the grid is (cell blocks x time
// blocks) instead of one thread walking 36 500 rows: float64 partial sums per block in ascending t, combined in
// ascending block order -- a fixed order, mirrored by the oracle.
template <int NC>  // NC == n_coef exactly: no per-term branch, and the NC doubles of a model row are ONE wide scalar load (with a
                   // runtime count every term was its own scalar load + wait: the passes were bound by scalar-load latency)
__global__ void __launch_bounds__(256)
k_detrend_partial(const float* __restrict__ x, long T, long C, const double* __restrict__ pmodel /*[T][n]*/, int n_coef,
                  double* __restrict__ partial /*[ntb][n][C]*/, int* __restrict__ invalid_count) {
    const long c = (long)blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    const long t0 = (long)blockIdx.y * DETREND_TBLOCK;
    const long t1 = t0 + DETREND_TBLOCK < T ? t0 + DETREND_TBLOCK : T;
    double acc[NC];
#pragma unroll
    for (int k = 0; k < NC; ++k) acc[k] = 0.0;
    int n_invalid = 0;
#pragma unroll DETREND_UNROLL
    for (long t = t0; t < t1; ++t) {
        const float v = x[(size_t)t * C + c];
        n_invalid += finite_f(v) ? 0 : 1;
        const double vd = (double)v;
        const double* pm = pmodel + (size_t)t * n_coef;
#pragma unroll
        for (int k = 0; k < NC; ++k) acc[k] += pm[k] * vd;
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

template <int NC, bool STORE>  // STORE false: only the sum is wanted (a branch on `out` inside the row loop keeps the loads of the
                               // unrolled rows from being issued together: 5.7 instead of 3.3 ms per band)
__global__ void __launch_bounds__(256)
k_detrend_resid(const float* __restrict__ x, long T, long C, const double* __restrict__ model_t /*[T][n]*/, int n_coef,
                const double* __restrict__ coef, float* __restrict__ out, double* __restrict__ psum /*[ntb][C]*/) {
    const long c = (long)blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    const long t0 = (long)blockIdx.y * DETREND_TBLOCK;
    const long t1 = t0 + DETREND_TBLOCK < T ? t0 + DETREND_TBLOCK : T;
    double cf[NC];
#pragma unroll
    for (int k = 0; k < NC; ++k) cf[k] = coef[(size_t)k * C + c];
    double sum = 0.0;
#pragma unroll DETREND_UNROLL
    for (long t = t0; t < t1; ++t) {
        const double* mt = model_t + (size_t)t * n_coef;
        double trend = 0.0;
#pragma unroll
        for (int k = 0; k < NC; ++k) trend += mt[k] * cf[k];
        const float r = x[(size_t)t * C + c] - (float)trend;
        if (STORE) out[(size_t)t * C + c] = r;
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

// coefs_only: no residual field is produced (out may be null) -- the fit, and with force_zero_mean the residual's mean into the
// context's scratch; *coef_dev / *mean_dev return where they are
static int detrend_impl(marex_ctx* ctx, const float* x, int64_t T, int64_t C, const double* pmodel, const double* model_t,
                        int n_coef, int force_zero_mean, float* out, uint8_t* mask, int32_t* invalid_count, float* mean_out,
                        bool coefs_only = false, const double** coef_dev = nullptr, const float** mean_dev = nullptr);

extern "C" int marex_detrend_f32(marex_ctx* ctx, const float* x, int64_t T, int64_t C, const double* pmodel,
                                 const double* model_t, int n_coef, int force_zero_mean, float* out, uint8_t* mask,
                                 int32_t* invalid_count) {
    return detrend_impl(ctx, x, T, C, pmodel, model_t, n_coef, force_zero_mean, out, mask, invalid_count, nullptr);
}

extern "C" int marex_detrend_deferred_mean_f32(marex_ctx* ctx, const float* x, int64_t T, int64_t C, const double* pmodel,
                                               const double* model_t, int n_coef, float* out, float* mean, uint8_t* mask,
                                               int32_t* invalid_count) {
    if (ctx && !mean) return fail(ctx, -1, "marex_detrend_deferred_mean_f32: null pointer");
    return detrend_impl(ctx, x, T, C, pmodel, model_t, n_coef, 1, out, mask, invalid_count, mean);
}

// detrend_fixed_baseline (detect.py:2400-2462) as one chain: fit (one read of x), residual mean (one read, nothing written),
// then the climatology kernel recomputes every residual from x while it reads its dayofyear bucket (one read, one write).
// Covers n_coef <= 5 and buckets <= 128 rows; -4 otherwise (callers then run the two stages separately).
static int tails32_check(marex_ctx* ctx, const char* who, const float* edges, int nb, void* lists, uint32_t* aux, int max_bucket,
                        int64_t C, int& nper) {
    if (!edges || !lists || !aux) return fail(ctx, -1, "%s: null pointer", who);
    if (nb < 4 || nb > TAIL_MAX_NB || max_bucket < 1 || max_bucket > TAIL_MAX_BUCKET || C > (1 << 24))
        return fail(ctx, -4, "%s: the key lists need nb <= %d, buckets of at most %d rows and C <= 2^24", who, TAIL_MAX_NB, TAIL_MAX_BUCKET);
    if (((uintptr_t)lists & 15) != 0) return fail(ctx, -1, "%s: lists must be 16-byte aligned", who);
    nper = (max_bucket + 31) / 32;
    return 0;
}

static int detrend_fixed_baseline_impl(marex_ctx* ctx, const float* x, int64_t T, int64_t C, const double* pmodel,
                                       const double* model_t, const double* model_sorted, int n_coef, int force_zero_mean,
                                       const int32_t* doy_start, const int32_t* doy_rows, const uint8_t* use_row, int max_bucket,
                                       float* out, uint8_t* mask, int32_t* invalid_count, const float* edges, int nb, void* lists,
                                       uint32_t* aux) {
    if (!ctx) return -1;
    if (!x || !pmodel || !model_t || !model_sorted || !doy_start || !doy_rows || !out || T <= 0 || C <= 0)
        return fail(ctx, -1, "marex_detrend_fixed_baseline_f32: null pointer or empty shape");
    if (n_coef < 1 || n_coef > 5 || max_bucket < 1 || max_bucket > 128)
        return fail(ctx, -4, "marex_detrend_fixed_baseline_f32: covers 1..5 coefficients and buckets of at most 128 rows");
    int nper = 0;
    if (lists) {
        const int rc = tails32_check(ctx, "marex_detrend_fixed_baseline_tails_f32", edges, nb, lists, aux, max_bucket, C, nper);
        if (rc) return rc;
    }
    const double* coef = nullptr;
    const float* mean = nullptr;
    const int rc = detrend_impl(ctx, x, T, C, pmodel, model_t, n_coef, force_zero_mean, nullptr, mask, invalid_count, nullptr, true,
                                &coef, &mean);
    if (rc != 0) return rc;
    const float* sub = force_zero_mean ? mean : nullptr;
    dim3 grid((unsigned)((C + 255) / 256), NDOY);
    uint4* tl = reinterpret_cast<uint4*>(lists);
    const size_t lds = lists ? (size_t)(nb + 1) * sizeof(float) : 0;
    {
        LaunchTimer lt(ctx, MAREX_K_FIXED);
#define MAREX_DF_LAUNCH(NM, N, TL)                                                                                                \
    hipLaunchKernelGGL((k_fixed_baseline_reg<NM, N, TL>), grid, dim3(256), lds, ctx->stream, x, (long)C, doy_start, doy_rows, use_row, \
                       out, nullptr, nullptr, sub, model_sorted, coef, edges, nb, tl, aux, nper)
#define MAREX_DF_CASE(N)                                                                                                          \
    case N:                                                                                                                       \
        if (max_bucket <= 48) {                                                                                                   \
            if (lists) MAREX_DF_LAUNCH(48, N, true); else MAREX_DF_LAUNCH(48, N, false);                                          \
        } else {                                                                                                                  \
            if (lists) MAREX_DF_LAUNCH(128, N, true); else MAREX_DF_LAUNCH(128, N, false);                                        \
        }                                                                                                                         \
        break;
        switch (n_coef) { MAREX_DF_CASE(1) MAREX_DF_CASE(2) MAREX_DF_CASE(3) MAREX_DF_CASE(4) MAREX_DF_CASE(5) }
#undef MAREX_DF_CASE
#undef MAREX_DF_LAUNCH
    }
    HIP_TRY(ctx, hipGetLastError());
    return 0;
}

extern "C" int marex_detrend_fixed_baseline_f32(marex_ctx* ctx, const float* x, int64_t T, int64_t C, const double* pmodel,
                                                const double* model_t, const double* model_sorted, int n_coef, int force_zero_mean,
                                                const int32_t* doy_start,
                                                const int32_t* doy_rows, const uint8_t* use_row, int max_bucket, float* out,
                                                uint8_t* mask, int32_t* invalid_count) {
    return detrend_fixed_baseline_impl(ctx, x, T, C, pmodel, model_t, model_sorted, n_coef, force_zero_mean, doy_start, doy_rows,
                                       use_row, max_bucket, out, mask, invalid_count, nullptr, 0, nullptr, nullptr);
}

// ... and the sorted key lists (32 rows per list) + aux words of the anomalies it writes: marex_tail_extract_f32(out) fused in
extern "C" int marex_detrend_fixed_baseline_tails_f32(marex_ctx* ctx, const float* x, int64_t T, int64_t C, const double* pmodel,
                                                      const double* model_t, const double* model_sorted, int n_coef,
                                                      int force_zero_mean, const int32_t* doy_start, const int32_t* doy_rows,
                                                      const uint8_t* use_row, int max_bucket, const float* edges, int nb, float* out,
                                                      uint8_t* mask, int32_t* invalid_count, void* lists, uint32_t* aux) {
    if (ctx && (!lists || !aux || !edges)) return fail(ctx, -1, "marex_detrend_fixed_baseline_tails_f32: null pointer");
    return detrend_fixed_baseline_impl(ctx, x, T, C, pmodel, model_t, model_sorted, n_coef, force_zero_mean, doy_start, doy_rows,
                                       use_row, max_bucket, out, mask, invalid_count, edges, nb, lists, aux);
}

// fixed_baseline (detect.py:2299-2397) with the key lists of its output: the register kernel (buckets of at most 128 rows)
extern "C" int marex_fixed_baseline_tails_f32(marex_ctx* ctx, const float* x, const float* sub, int max_bucket, int64_t T, int64_t C,
                                              const int32_t* doy_start, const int32_t* doy_rows, const uint8_t* use_row,
                                              const float* edges, int nb, float* out, uint8_t* mask, int32_t* invalid_count,
                                              void* lists, uint32_t* aux) {
    if (!ctx) return -1;
    if (!x || !doy_start || !doy_rows || !out || T <= 0 || C <= 0)
        return fail(ctx, -1, "marex_fixed_baseline_tails_f32: null pointer or empty shape");
    int nper = 0;
    const int rc = tails32_check(ctx, "marex_fixed_baseline_tails_f32", edges, nb, lists, aux, max_bucket, C, nper);
    if (rc) return rc;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    dim3 grid((unsigned)((C + 255) / 256), NDOY);
    const size_t lds = (size_t)(nb + 1) * sizeof(float);
    {
        LaunchTimer lt(ctx, MAREX_K_FIXED);
        if (max_bucket <= 48)
            hipLaunchKernelGGL((k_fixed_baseline_reg<48, 0, true>), grid, dim3(256), lds, ctx->stream, x, (long)C, doy_start, doy_rows,
                               use_row, out, mask, invalid_count, sub, nullptr, nullptr, edges, nb, reinterpret_cast<uint4*>(lists), aux, nper);
        else
            hipLaunchKernelGGL((k_fixed_baseline_reg<128, 0, true>), grid, dim3(256), lds, ctx->stream, x, (long)C, doy_start, doy_rows,
                               use_row, out, mask, invalid_count, sub, nullptr, nullptr, edges, nb, reinterpret_cast<uint4*>(lists), aux, nper);
    }
    HIP_TRY(ctx, hipGetLastError());
    return 0;
}

static int detrend_impl(marex_ctx* ctx, const float* x, int64_t T, int64_t C, const double* pmodel, const double* model_t,
                        int n_coef, int force_zero_mean, float* out, uint8_t* mask, int32_t* invalid_count, float* mean_out,
                        bool coefs_only, const double** coef_dev, const float** mean_dev) {
    if (!ctx) return -1;
    if (!x || !pmodel || !model_t || (!out && !coefs_only) || T <= 0 || C <= 0)
        return fail(ctx, -1, "marex_detrend_f32: null pointer or empty shape");
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
    if (coef_dev) *coef_dev = coef;
    if (mean_dev) *mean_dev = mean;
    if (coefs_only) {
        out = nullptr;
        mean_out = mean;
    }
    const unsigned ncb = (unsigned)((C + 255) / 256);
    {
        LaunchTimer lt(ctx, MAREX_K_DETREND);
#define MAREX_DETREND_CASE(N)                                                                                                  \
    case N:                                                                                                                    \
        hipLaunchKernelGGL(k_detrend_partial<N>, dim3(ncb, ntb), dim3(256), 0, ctx->stream, x, (long)T, (long)C, pmodel, n_coef, \
                           partial, invalid_count);                                                                            \
        break;
        switch (n_coef) {
            MAREX_DETREND_CASE(1) MAREX_DETREND_CASE(2) MAREX_DETREND_CASE(3) MAREX_DETREND_CASE(4) MAREX_DETREND_CASE(5)
            MAREX_DETREND_CASE(6) MAREX_DETREND_CASE(7) MAREX_DETREND_CASE(8) MAREX_DETREND_CASE(9) MAREX_DETREND_CASE(10)
            MAREX_DETREND_CASE(11) MAREX_DETREND_CASE(12)
        }
#undef MAREX_DETREND_CASE
        hipLaunchKernelGGL(k_detrend_combine, dim3(ncb), dim3(256), 0, ctx->stream, x, (long)C, ntb, n_coef, partial, coef,
                           mask);
#define MAREX_DETREND_CASE(N)                                                                                                  \
    case N:                                                                                                                    \
        if (out)                                                                                                               \
            hipLaunchKernelGGL((k_detrend_resid<N, true>), dim3(ncb, ntb), dim3(256), 0, ctx->stream, x, (long)T, (long)C, model_t, \
                               n_coef, coef, out, partial);                                                                    \
        else                                                                                                                   \
            hipLaunchKernelGGL((k_detrend_resid<N, false>), dim3(ncb, ntb), dim3(256), 0, ctx->stream, x, (long)T, (long)C, model_t, \
                               n_coef, coef, out, partial);                                                                    \
        break;
        if (!coefs_only || force_zero_mean)  // coefficients only and no mean wanted: the residual pass has nothing to do
        switch (n_coef) {
            MAREX_DETREND_CASE(1) MAREX_DETREND_CASE(2) MAREX_DETREND_CASE(3) MAREX_DETREND_CASE(4) MAREX_DETREND_CASE(5)
            MAREX_DETREND_CASE(6) MAREX_DETREND_CASE(7) MAREX_DETREND_CASE(8) MAREX_DETREND_CASE(9) MAREX_DETREND_CASE(10)
            MAREX_DETREND_CASE(11) MAREX_DETREND_CASE(12)
        }
#undef MAREX_DETREND_CASE
        if (force_zero_mean && mean_out) {  // the caller subtracts (marex_fixed_baseline_sub_f32 does it on load)
            hipLaunchKernelGGL(k_detrend_mean, dim3(ncb), dim3(256), 0, ctx->stream, (long)T, (long)C, ntb, partial, mean_out);
        } else if (force_zero_mean) {
            hipLaunchKernelGGL(k_detrend_mean, dim3(ncb), dim3(256), 0, ctx->stream, (long)T, (long)C, ntb, partial, mean);
            hipLaunchKernelGGL(k_detrend_sub, dim3((unsigned)((C / 4 + 256) / 256), (unsigned)((T + 63) / 64)), dim3(256), 0,
                               ctx->stream, (long)T, (long)C, mean, out);
        }
    }
    HIP_TRY(ctx, hipGetLastError());
    return 0;
}

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
