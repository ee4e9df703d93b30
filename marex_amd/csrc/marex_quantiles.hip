// marex_quantiles.hip -- exact Hobday percentile and global (constant-in-time) thresholds
#include "marex_common.hip.h"

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
    const bool small_counts = T_out <= 65535 && ctx_opt(ctx, "GLOBAL_V1", 0) == 0;  // uint16 counters suffice
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
