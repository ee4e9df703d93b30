// marex_mask.hip -- extreme mask kernels and the threshold transpose
#include "marex_common.hip.h"

// ------------------------------------------------------------------------------------------------
// K_M: extreme[t, c] = anom[t, c] >= thr[doy(t), c]
// One workgroup = (VEC*256 cells, a chunk of consecutive dayofyears).  Rows are visited grouped by
// dayofyear so that one threshold row serves all its timesteps out of registers; 16-byte loads of the
// anomaly row, 4-byte stores of the mask, MASK_UNROLL independent rows in flight per lane.
// ------------------------------------------------------------------------------------------------
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

// The same mask from the BIN matrix of the anomaly kernel (2 bytes per sample instead of 4): with k = bin(anom) and
// kt = bin(thr) on the same increasing edge table, k > kt implies anom >= thr and k < kt implies anom < thr; only samples
// in the threshold's own bin -- and those in the overflow bin nb, which also holds NaN -- need the anomaly itself
// (about one sample in 500 for 0.01-wide bins).  A NaN threshold (land) never matches.  Bit-identical to k_mask_ge.
template <int VEC>  // cells per thread: 4 or 8 (8-/16-byte bin loads, 4-/8-byte mask stores)
__global__ void __launch_bounds__(256)
k_mask_ge_bins(const float* __restrict__ anom, const unsigned short* __restrict__ bins, const float* __restrict__ edges,
               int nb, const float* __restrict__ thr, const int* __restrict__ doy_start, const int* __restrict__ doy_rows,
               long T_out, long C, long c0, long c1, unsigned char* __restrict__ out, unsigned long long* __restrict__ n_true) {
    constexpr int NW = VEC / 2;  // dwords of bins per row
    typedef unsigned bw_t __attribute__((ext_vector_type(NW)));
    const int nchunk = (int)gridDim.y;
    const int dA = (int)blockIdx.y * NDOY / nchunk, dB = ((int)blockIdx.y + 1) * NDOY / nchunk;
    const long c = c0 + ((long)blockIdx.x * 256 + threadIdx.x) * VEC;
    unsigned cnt = 0;
    if (c < c1) {
        const float inv_width = (float)(nb - 1) / (edges[nb] - edges[1]);
        const unsigned short* bcol = bins + bins_index(0, c, T_out);  // VEC cells of one 16-cell block; rows 16 elements apart
        for (int d = dA; d < dB; ++d) {
            const int r0 = doy_start[d], r1 = doy_start[d + 1];
            if (r0 == r1) continue;
            float tv[VEC];
            int kt[VEC];
#pragma unroll
            for (int q = 0; q < VEC / 4; ++q) {
                const float4 th = *reinterpret_cast<const float4*>(thr + (size_t)d * C + c + 4 * q);
                tv[4 * q] = th.x, tv[4 * q + 1] = th.y, tv[4 * q + 2] = th.z, tv[4 * q + 3] = th.w;
            }
#pragma unroll
            for (int i = 0; i < VEC; ++i) kt[i] = (tv[i] == tv[i]) ? digitize_bin(tv[i], edges, nb, inv_width) : 0x7fff;
            auto one_row = [&](int r, bw_t b) {
                const size_t off = (size_t)doy_rows[r] * C + c;
                unsigned m[VEC];
                bool need = false;
#pragma unroll
                for (int i = 0; i < VEC; ++i) {
                    const int k = (int)((b[i >> 1] >> ((i & 1) * 16)) & 0xFFFFu);
                    m[i] = k > kt[i];
                    need = need || k == kt[i] || (k == nb && kt[i] != 0x7fff);
                }
                if (need) {  // rare: settle the cells on the values themselves
#pragma unroll
                    for (int q = 0; q < VEC / 4; ++q) {
                        const float4 a = *reinterpret_cast<const float4*>(anom + off + 4 * q);
                        m[4 * q] = a.x >= tv[4 * q];
                        m[4 * q + 1] = a.y >= tv[4 * q + 1];
                        m[4 * q + 2] = a.z >= tv[4 * q + 2];
                        m[4 * q + 3] = a.w >= tv[4 * q + 3];
                    }
                }
                typedef unsigned mw_t __attribute__((ext_vector_type(VEC / 4)));
                mw_t w;
#pragma unroll
                for (int q = 0; q < VEC / 4; ++q) {
                    w[q] = m[4 * q] | (m[4 * q + 1] << 8) | (m[4 * q + 2] << 16) | (m[4 * q + 3] << 24);
                    cnt += m[4 * q] + m[4 * q + 1] + m[4 * q + 2] + m[4 * q + 3];
                }
                if constexpr (VEC == 4)
                    __builtin_nontemporal_store(w[0], reinterpret_cast<unsigned*>(out + off));
                else
                    __builtin_nontemporal_store(w, reinterpret_cast<mw_t*>(out + off));
            };
            auto load_row = [&](int r) { return *reinterpret_cast<const bw_t*>(bcol + (size_t)r * 16); };
            int r = r0;
            for (; r + 4 <= r1; r += 4) {
                bw_t b[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) b[u] = load_row(r + u);
#pragma unroll
                for (int u = 0; u < 4; ++u) one_row(r + u, b[u]);
            }
            for (; r < r1; ++r) one_row(r, load_row(r));
        }
    }
    if (n_true) {
        for (int s = 32; s > 0; s >>= 1) cnt += __shfl_down(cnt, s, 64);
        if ((threadIdx.x & 63) == 0 && cnt) atomicAdd(n_true, (unsigned long long)cnt);
    }
}

extern "C" int marex_mask_ge_doy_bins_f32(marex_ctx* ctx, const float* anom, const uint16_t* bins, const float* edges, int nb,
                                          const float* thr_doy_major, const int32_t* doy_start, const int32_t* doy_rows,
                                          int64_t T_out, int64_t C, int64_t c0, int64_t c1, uint8_t* extreme,
                                          unsigned long long* n_true) {
    if (!ctx) return -1;
    if (!anom || !bins || !edges || !thr_doy_major || !doy_start || !doy_rows || !extreme || T_out <= 0 || C <= 0 || nb < 4)
        return fail(ctx, -1, "marex_mask_ge_doy_bins_f32: null pointer or empty shape");
    if (c0 < 0 || c1 > C || c0 >= c1) return fail(ctx, -1, "marex_mask_ge_doy_bins_f32: need 0 <= c0 < c1 <= C");
    const bool vec = (C % 4 == 0) && (c0 % 4 == 0) && (c1 % 4 == 0) && (((uintptr_t)anom | (uintptr_t)thr_doy_major) % 16 == 0) &&
                     ((uintptr_t)extreme % 4 == 0) && ((uintptr_t)bins % 8 == 0) && nb < 0x7fff;
    // Short dayofyear buckets (few years) make the per-bucket runs of the bin matrix shorter than a cache line or two: the
    // plain compare is faster there (10-yr field: 1.9 vs 2.1 ms); long ones win by a third (100-yr band: 4.2 -> 2.9 ms).
    const int mode = ctx_opt(ctx, "MASK_BINS", -1);  // -1 auto, 0 never, 1 whenever the shapes allow
    const bool worth = mode == 1 || (mode < 0 && T_out / NDOY >= 24);
    if (!vec || !worth)  // also: shapes the 4-cell kernel does not cover
        return marex_mask_ge_doy_f32(ctx, anom, thr_doy_major, doy_start, doy_rows, T_out, C, c0, c1, extreme, n_true);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    {
        LaunchTimer lt(ctx, MAREX_K_MASK);
        const bool wide = (C % 8 == 0) && (c0 % 8 == 0) && (c1 % 8 == 0) && ((uintptr_t)bins % 16 == 0) &&
                          ((uintptr_t)extreme % 8 == 0) && ctx_opt(ctx, "MASK_VEC", 8) == 8;
        const int vecw = wide ? 8 : 4;
        const unsigned ncb = (unsigned)(((c1 - c0) / vecw + 255) / 256);
        unsigned chunks = (4096 + ncb - 1) / ncb;
        chunks = chunks < MASK_DOY_CHUNKS ? MASK_DOY_CHUNKS : (chunks > 61 ? 61 : chunks);
        if (wide)
            hipLaunchKernelGGL(k_mask_ge_bins<8>, dim3(ncb, chunks), dim3(256), 0, ctx->stream, anom, bins, edges, nb,
                               thr_doy_major, doy_start, doy_rows, (long)T_out, (long)C, (long)c0, (long)c1, extreme, n_true);
        else
            hipLaunchKernelGGL(k_mask_ge_bins<4>, dim3(ncb, chunks), dim3(256), 0, ctx->stream, anom, bins, edges, nb,
                               thr_doy_major, doy_start, doy_rows, (long)T_out, (long)C, (long)c0, (long)c1, extreme, n_true);
    }
    HIP_TRY(ctx, hipGetLastError());
    return 0;
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
            {
                const int forced = ctx_opt(ctx, "MASK_CHUNKS", 0);  // dayofyear chunks of the grid (experiments)
                if (forced >= 1 && forced <= 366) chunks = (unsigned)forced;
            }
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

// Verdict of the validation (detect.py:205-279) from the per-cell results of the anomaly kernels: over cells c0 .. c1-1
// out[0] = ocean cells (mask != 0), out[1] = non-finite values in ocean cells, out[2] = ocean cells with any,
// out[3] = the largest count in one ocean cell.
__global__ void __launch_bounds__(256)
k_validation_summary(const unsigned char* __restrict__ mask, const int* __restrict__ invalid, long c0, long c1,
                     long long* __restrict__ out) {
    long long n_ocean = 0, total = 0, cells = 0;
    int worst = 0;
    for (long c = c0 + (long)blockIdx.x * blockDim.x + threadIdx.x; c < c1; c += (long)gridDim.x * blockDim.x) {
        const int m = mask[c] ? 1 : 0;
        const int inv = m ? invalid[c] : 0;
        n_ocean += m;
        total += inv;
        cells += inv > 0;
        worst = inv > worst ? inv : worst;
    }
    for (int sft = 32; sft > 0; sft >>= 1) {
        n_ocean += __shfl_down(n_ocean, sft, 64);
        total += __shfl_down(total, sft, 64);
        cells += __shfl_down(cells, sft, 64);
        const int w = __shfl_down(worst, sft, 64);
        worst = w > worst ? w : worst;
    }
    if ((threadIdx.x & 63) == 0) {
        if (n_ocean) atomicAdd((unsigned long long*)&out[0], (unsigned long long)n_ocean);
        if (total) atomicAdd((unsigned long long*)&out[1], (unsigned long long)total);
        if (cells) atomicAdd((unsigned long long*)&out[2], (unsigned long long)cells);
        if (worst) atomicMax(&out[3], (long long)worst);
    }
}

extern "C" int marex_validation_summary(marex_ctx* ctx, const uint8_t* mask, const int32_t* invalid_count, int64_t c0,
                                        int64_t c1, int64_t* out4) {
    if (!ctx) return -1;
    if (!mask || !invalid_count || !out4 || c0 < 0 || c1 < c0)
        return fail(ctx, -1, "marex_validation_summary: null pointer or bad cell range");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipMemsetAsync(out4, 0, 4 * sizeof(int64_t), ctx->stream));
    if (c1 > c0) {
        const long n = (long)(c1 - c0);
        const int blocks = (int)((n + 1023) / 1024 < 1024 ? (n + 1023) / 1024 : 1024);
        hipLaunchKernelGGL(k_validation_summary, dim3(blocks), dim3(256), 0, ctx->stream, mask, invalid_count, (long)c0,
                           (long)c1, reinterpret_cast<long long*>(out4));
        HIP_TRY(ctx, hipGetLastError());
    }
    return 0;
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
