// marex_shifting.hip -- K_A: validation + smoothing + rolling climatology + anomaly + bins
#include "marex_common.hip.h"
#include "marex_tails.hip.h"

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
#ifndef MAREX_ABLATION
    ablate = 0;  // the timing-only ablation bits (wrong results by design) exist only in -DMAREX_ABLATION builds
#endif
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

// 16-byte list stores: descriptor words built by hand (base, no stride, unbounded, raw dword format) so that the store can
// be issued from inline assembly TOGETHER with the wait states it needs.  The hazard is the ISA manual's "VMEM store of more
// than 64 bits of data followed by a VALU write of its vdata VGPRs" row of the manual-wait-state table (the programming
// guide's rule for hand-issued dwordx3 / dwordx4 stores: end the asm string with s_nop 1 = two wait states,
// cdna_hip_programming.md 5.7 item 1).  LLVM's hazard recogniser pads that row ONLY when soffset is not a register
// (GCNHazardRecognizer::createsVALUHazard exempts MUBUF stores with an SGPR soffset), so for this store -- SGPR soffset, data
// registers rewritten by the very next VALU instruction of the sort network -- nothing was inserted, and a separate s_nop
// statement was scheduled away; observed on gfx950 / ROCm 7.2: the first dword of a chunk lost in lanes 12-15 of every 16.
// The store and its wait states are therefore ONE asm statement: s_nop 2 = three wait states, one more than the table's two
// (the row is documented for the no-soffset form; the extra state costs 4 cycles per 1-KiB store).
typedef int tl_out_rsrc_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ tl_out_rsrc_t tl_out_make_rsrc(const void* base) {
    const unsigned long long b = (unsigned long long)base;
    tl_out_rsrc_t r;
    r.x = __builtin_amdgcn_readfirstlane((int)(unsigned)b);
    r.y = __builtin_amdgcn_readfirstlane((int)((unsigned)(b >> 32) & 0xFFFFu));
    r.z = -1;
    r.w = 0x00020000;
    return r;
}
__device__ __forceinline__ void tl_out_store(tl_out_rsrc_t r, unsigned voff, unsigned soff, const unsigned (&w)[4]) {
    typedef unsigned u4 __attribute__((ext_vector_type(4)));
    const u4 v = {w[0], w[1], w[2], w[3]};
    const int soff_u = __builtin_amdgcn_readfirstlane((int)soff);
    asm volatile("buffer_store_dwordx4 %0, %1, %2, %3 offen\n\ts_nop 2" : : "v"(v), "v"(voff), "s"(r), "s"(soff_u) : "memory");
}

#define SHIFT_INFO_WORDS 128  // [0..91] chunk handled by the fast kernel, [92] arange edges usable
#define CLASSIFY_SPLIT 11      // threads per chunk in k_shift_classify (92 * 11 = 1012 <= 1024)

// fplan[year][chunk] = two int4: {timestep of the chunk's first dayofyear (-1: absent), its output row (-1: none), number of
// leading dayofyears present (0..4), bin-matrix row of dayofyear 0} and {bin-matrix rows of dayofyears 1..3, timestep of the
// first dayofyear of the chunk's WORKGROUP (16 dayofyears) in the FOLLOWING year}: everything k_shift_fast needs per year
// in ONE 32-byte scalar load, every field live (a dead field lets the register allocator recycle its SGPR and wait for
// the load on the spot)
__global__ void k_shift_classify(const int4* __restrict__ year_plan, int n_cal, const float* __restrict__ edges, int nb,
                                 int want_bins, int enable, int* __restrict__ info, int4* __restrict__ fplan) {
    __shared__ int s_edges_ok;
    const int t = threadIdx.x;
    int eok = 1;
    if (want_bins) {  // the table is an arange: every thread checks its share of the entries
        const float first = edges[1], delta = edges[2] - edges[1];
        for (int j = 1 + t; j <= nb; j += (int)blockDim.x)
            eok = eok && __float_as_uint(edges[j]) == __float_as_uint(arange_edge(j, first, delta));
    }
    eok = __syncthreads_and(eok);
    if (t == 0) {
        int ok = eok;
        if (want_bins) {
            const float first = edges[1], delta = edges[2] - edges[1], last = edges[nb];
            ok = ok && delta > 0.f && nb < 32768;
            // the fused guess (biased down by 1/128 bin) must land in the true bin or the one below: generous bound
            // on its rounding error (about 8x what the individual roundings add up to)
            const double m = fabs((double)first) > fabs((double)last) ? fabs((double)first) : fabs((double)last);
            ok = ok && ((double)nb + 2.0 * m / (double)delta) * (1.0 / 1048576.0) < 1.0 / 256.0;
        }
        s_edges_ok = ok && enable;
        info[92] = s_edges_ok;
    }
    __shared__ int s_ok[92];
    if (t < 92) s_ok[t] = 1;
    __syncthreads();
    // every year: the first m (0..4) dayofyears of the chunk present on consecutive timesteps, the rest absent
    // (leap day; dayofyears past 366 in the last chunk); output rows all or none, consecutive.
    // CLASSIFY_SPLIT threads share the years of one chunk (the checks are chains of dependent global loads).
    const int chunk = t / CLASSIFY_SPLIT, sub = t % CLASSIFY_SPLIT;
    if (chunk < 92) {
        const int d0 = chunk * 4;
        int ok = 1;
        for (int y = sub; y < n_cal; y += CLASSIFY_SPLIT) {
            int4 e[4];
            for (int i = 0; i < 4; ++i)
                e[i] = (d0 + i < NDOY) ? year_plan[(size_t)y * NDOY + d0 + i] : make_int4(-1, -1, -1, 0);
            int m = 0;
            while (m < 4 && e[m].x >= 0) ++m;
            if (fplan) {
                const int yn = y + 1 < n_cal ? y + 1 : n_cal - 1;
                const int tbw = year_plan[(size_t)yn * NDOY + (chunk >> 2) * 16].x;
                fplan[((size_t)y * 92 + chunk) * 2] = make_int4(e[0].x, e[0].y, m, e[0].z);
                fplan[((size_t)y * 92 + chunk) * 2 + 1] = make_int4(e[1].z, e[2].z, e[3].z, tbw);
            }
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
        if (!ok) atomicAnd(&s_ok[chunk], 0);
    }
    __syncthreads();
    if (t < 92) info[t] = s_ok[t] && s_edges_ok;
}

// TAILS: instead of the bin matrix the kernel emits the sorted key lists of marex_tails.hip.h.  The SHIFT_LIST (15) most recent output
// years of the wave's 4 dayofyears wait as packed key pairs in LDS (one uniform slot per year, no per-lane counters);
// every 15th year the wave sorts them (63 packed compare-exchanges per pair of dayofyears) and writes one list per
// dayofyear as two 16-byte chunks per lane -- whole 1-KiB lines per wave, nothing is ever read back.
#ifndef SHIFT_EXP
#define SHIFT_EXP 0  // timing experiments (wrong results by design), alt builds only: 1 no barriers, 2 no row prefetch, 4 no anomaly stores, 8 no keys, 16 no smoothing sums, 32 no climatology sums
#endif
#define SHIFT_LIST 15   // output years per emitted list (15 x 2 pairs x 256 B x 4 waves + one 9-KiB stage = 39 KiB: 4 workgroups per CU)
struct TailOut {
    uint4* lists;              // [366][NPER][2][C] chunks
    unsigned short* aux;       // [366][C]
    const int* doy_start;      // [367] first bin-matrix row of every dayofyear (key positions = row - doy_start)
    int nper;
    unsigned long long* dbg;   // debug counters (-DSHIFT_STAMPS builds: phase timers)
};

template <int W, bool TAILS, int S = 21>
__global__ void __launch_bounds__(256)
k_shift_fast(const float* __restrict__ x, long T, long C, const int4* __restrict__ fplan, int n_cal,
             const int* __restrict__ info, int write_clim, const float* __restrict__ edges, int nb, long T_out, float* __restrict__ out, unsigned short* __restrict__ bins, unsigned char* __restrict__ mask,
             int* __restrict__ invalid_count, int ncg, int nblk, TailOut tails) {
    int cg, bc;
    if (!xcd_swizzle(blockIdx.x, ncg, nblk, cg, bc)) return;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int lane = (int)(threadIdx.x & 63);
    const int chunk = bc * 4 + wave;
    // The 4 waves work on 4 neighbouring chunks of the same 64 cells: the 36 rows one calendar year needs for
    // all 16 dayofyears are staged once in LDS (double buffered, loaded one year ahead) and every wave reads its
    // 24 from there -- 2.25 instead of 6 row reads per output row leave the L2.
    // TAILS: ONE stage buffer (a second barrier per year separates its readers from the next year's writers) so that the
    // waiting keys fit beside it at four workgroups per CU
    // smoothing width S (odd): a wave needs S + 3 rows for its 4 dayofyears, the workgroup S + 15, H = S / 2 before the first
    static_assert((S & 1) == 1 && S >= 5 && S <= 25, "k_shift_fast: odd smoothing widths 5..25");
    constexpr int H = S / 2, NPAIR = (S + 3) / 2, NST = S + 15, RPW = (NST + 3) / 4;  // RPW: rows each wave stages per year
    constexpr int NSTAGE = TAILS ? 1 : 2;
    __shared__ float stage[NSTAGE][NST * 64];
    __shared__ unsigned newkeys[TAILS ? 4 : 1][2][TAILS ? SHIFT_LIST : 1][64];  // [wave][pair of dayofyears][year slot][lane]
#ifdef SHIFT_PAD  // experiment: extra LDS (floats) to lower the number of resident workgroups
    __shared__ float lds_pad[SHIFT_PAD];
    if (T == -12345) out[0] = lds_pad[threadIdx.x];
#endif
    const bool mine = chunk < 92 && info[chunk] != 0;  // wave-uniform; the other waves only help staging
    const int d0 = mine ? chunk * 4 : 0;
    // per-year records (k_shift_classify): the wave's chunk, and the workgroup's first chunk (its first timestep places the
    // staged rows)
    const int4* prec = fplan + (size_t)chunk * 2;  // chunk < 92: 23 workgroups of 4
    constexpr size_t YREC = 92 * 2;  // int4 per year
    const long c = (long)cg * 64 + lane;
    const bool active = c < C;
    const unsigned cidx = active ? (unsigned)c : (unsigned)(C - 1);  // lanes beyond C duplicate the last cell
    const bool do_bins = bins != nullptr && !TAILS;
    // bin matrix: lane part of the element index relative to the wave's first 16-cell block
    const unsigned voff = cidx * 4u;  // byte offset of the lane's cell inside a (time, cell) row
    const int rowb = (int)(C * 4);    // bytes per (time, cell) row
    const unsigned bin_lane = (((cidx >> 4) - (unsigned)(cg * 4)) * (unsigned)T_out * 16u + (cidx & 15u)) * 2u;  // bytes
    const rsrc_t rbins = make_rsrc(do_bins ? bins + (size_t)(cg * 4) * (size_t)T_out * 16 : nullptr);

    float e_first = 0.f, e_delta = 1.f, inv_width = 1.f, e_last = 0.f;
    if (do_bins || TAILS) {
        e_first = edges[1];
        e_delta = edges[2] - edges[1];
        e_last = edges[nb];
        inv_width = (float)(nb - 1) / (e_last - e_first);
    }
    constexpr float Sf = (float)S;
    const float yS = 1.0f / Sf;
    const float Wf = (float)W;
    const float yW = 1.0f / Wf;
    const float nbm1f = (float)(nb - 1);
    const float c0 = (1.0f - e_first * inv_width) - 0.0078125f;  // +1 (edges[0] = -inf) and the 1/128-bin downward bias
    const float qnan = nan_f();

    if (chunk == 0 && mine && mask && active) mask[c] = finite_f(x[c]) ? 1 : 0;
    // rows tb-H .. tb+15+H (tb = timestep of the workgroup's first dayofyear in that year) can be staged when they
    // all lie inside the series; this wave loads rows RPW*wave .. RPW*wave+RPW-1 of them (the last wave fewer when 4 RPW > NST)
    auto stage_ok = [&](int tb) { return tb >= H && (long)tb + 16 + H <= T; };
    auto stage_load = [&](int tb, float (&nx)[RPW]) {
        const rsrc_t rs = make_rsrc(x + (size_t)(tb - H + RPW * wave) * C);
#pragma unroll
        for (int k = 0; k < RPW; ++k)
            if (4 * RPW == NST || RPW * wave + k < NST) nx[k] = ldb_f32(rs, voff, k * rowb);
    };
    auto stage_store = [&](int buf, const float (&nx)[RPW]) {
#pragma unroll
        for (int k = 0; k < RPW; ++k)
            if (4 * RPW == NST || RPW * wave + k < NST) stage[buf & (NSTAGE - 1)][(RPW * wave + k) * 64 + lane] = nx[k];
    };
    int tb_cur = fplan[(size_t)(bc * 4) * 2].x;
    int tb_n1 = prec[1].w;  // year 1 (record of year 0), known one iteration ahead of its row prefetch
    {
        float nx[RPW];
        if (stage_ok(tb_cur)) {
            stage_load(tb_cur, nx);
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
    // tails: valid-key counts and "value beyond the table" flags of the 4 dayofyears (packed pairs), year slot, list index
    unsigned t_cntA = 0, t_cntB = 0, t_ovfA = 0, t_ovfB = 0;
    int t_slot = 0, t_list = 0;
    int t_ds0 = 0, t_ds1 = 0, t_ds2 = 0, t_ds3 = 0;
    if (TAILS && mine) {
        t_ds0 = tails.doy_start[d0];
        t_ds1 = tails.doy_start[d0 + 1 < NDOY ? d0 + 1 : NDOY - 1];
        t_ds2 = tails.doy_start[d0 + 2 < NDOY ? d0 + 2 : NDOY - 1];
        t_ds3 = tails.doy_start[d0 + 3 < NDOY ? d0 + 3 : NDOY - 1];
    }
    // sort the waiting years of both pairs and write them out as list `t_list` of the wave's dayofyears
    auto tails_flush = [&]() {
        const tl_out_rsrc_t rl = tl_out_make_rsrc(tails.lists + (size_t)d0 * tails.nper * 2 * (size_t)C);
        const unsigned lvoff = cidx * 16u;
#pragma unroll
        for (int pr = 0; pr < 2; ++pr) {
            unsigned v[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) v[u] = (u < SHIFT_LIST && u < t_slot) ? newkeys[wave][pr][u][lane] : 0u;  // uniform bound
            sort16_desc(v);
#pragma unroll
            for (int h = 0; h < 2; ++h) {  // low / high halves = the pair's first / second dayofyear
                const int di = 2 * pr + h;
                if (d0 + di < NDOY) {  // uniform (the last chunk has two dayofyears that do not exist)
#pragma unroll
                    for (int jj = 0; jj < 2; ++jj) {
                        unsigned w[4];
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const unsigned a = v[8 * jj + 2 * i], b = v[8 * jj + 2 * i + 1];
                            w[i] = h == 0 ? ((a & 0xFFFFu) | (b << 16)) : ((a >> 16) | (b & 0xFFFF0000u));
                        }
                        const int li = __builtin_amdgcn_readfirstlane((di * tails.nper + t_list) * 2 + jj);
                        tl_out_store(rl, lvoff, (unsigned)li * (unsigned)C * 16u, w);
                    }
                }
            }
        }
        ++t_list;
        t_slot = 0;
    };

    int4 nA = prec[0], nB = prec[1];  // year 0
#ifdef SHIFT_STAMPS
    unsigned long long st_top = 0, st_mid = 0, st_end = 0, st_flush = 0;
#define SSTAMP(v) const unsigned long long v = __builtin_amdgcn_s_memtime()
#else
#define SSTAMP(v)
#endif
    auto one_year = [&](int y, auto Jc) {
        constexpr int J = decltype(Jc)::value;
        SSTAMP(tsA);
        const int4 pA = nA, pB = nB;  // {first timestep, first output row, dayofyears present, row 0}, {rows 1..3, next tb}
        const int tb = tb_cur;
        float nx[RPW];
        // next year's rows, one iteration ahead (its first timestep arrived during the previous iteration)
        const bool stage_next = y + 1 < n_cal && stage_ok(tb_n1) && !(SHIFT_EXP & 2);
        if (stage_next) stage_load(tb_n1, nx);
        v2f smA = splat2(qnan), smB = splat2(qnan);
        v2f xp[NPAIR];  // xp[m] = rows (r0 + 2m, r0 + 2m + 1), r0 = first timestep - H
        const bool staged = mine && pA.z > 0 && stage_ok(tb) && pA.x == tb + 4 * wave;
        if (staged) {
            const float* st = &stage[y & (NSTAGE - 1)][(4 * wave) * 64 + lane];
#pragma unroll
            for (int m = 0; m < NPAIR; ++m) {
                xp[m].x = st[(2 * m) * 64];
                xp[m].y = st[(2 * m + 1) * 64];
            }
        }
        if (TAILS) {
            // single stage buffer: every wave has its rows in registers before anyone overwrites the buffer with the next
            // year's (LDS traffic only: no vector-memory wait here, the row prefetch stays in flight)
            if (!(SHIFT_EXP & 1)) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        }
        else asm volatile("" ::: "memory");
        // scalar loads of the years to come go out HERE, after the wait for the staged rows: they are in flight during the
        // arithmetic instead of in front of three dependent waits at the top of the loop
        {
            const int y1 = y + 1 < n_cal ? y + 1 : n_cal - 1;
            nA = prec[(size_t)y1 * YREC];
            nB = prec[(size_t)y1 * YREC + 1];
            tb_cur = tb_n1;
            tb_n1 = nB.w;  // first timestep of the workgroup in year y + 2
        }
        SSTAMP(tsB);
        if (mine && pA.z > 0) {
            const long r0 = (long)pA.x - H;
            const bool edge = r0 < 0 || r0 + 2 * NPAIR > T;
            if (staged) {
            } else if (!edge) {
                const rsrc_t rx = make_rsrc(x + (size_t)r0 * C);
#pragma unroll
                for (int m = 0; m < NPAIR; ++m) {
                    xp[m].x = ldb_f32(rx, voff, (2 * m) * rowb);
                    xp[m].y = ldb_f32(rx, voff, (2 * m + 1) * rowb);
                }
            } else {
#pragma unroll
                for (int m = 0; m < NPAIR; ++m) {
                    long ra = r0 + 2 * m, rb = ra + 1;
                    ra = ra < 0 ? 0 : (ra > T - 1 ? T - 1 : ra);
                    rb = rb < 0 ? 0 : (rb > T - 1 ? T - 1 : rb);
                    xp[m].x = ldb_f32(make_rsrc(x + (size_t)ra * C), voff, 0);
                    xp[m].y = ldb_f32(make_rsrc(x + (size_t)rb * C), voff, 0);
                }
            }
            // smoothing: sequential sums of rows i .. i+S-1 for the four dayofyears i = 0..3 (S odd: rows S and S+2 are the
            // high halves of their pairs)
            v2f accA = (v2f){xp[0].x, -0.0f};
#pragma unroll
            for (int s = 1; s <= ((SHIFT_EXP & 16) ? 2 : S - 1); ++s) accA = (s & 1) ? pk_add_bc_hi(accA, xp[s >> 1]) : pk_add_bc_lo(accA, xp[s >> 1]);
            accA.y += xp[S >> 1].y;
            v2f accB = (v2f){xp[1].x, -0.0f};
#pragma unroll
            for (int s = 3; s <= ((SHIFT_EXP & 16) ? 4 : S + 1); ++s) accB = (s & 1) ? pk_add_bc_hi(accB, xp[s >> 1]) : pk_add_bc_lo(accB, xp[s >> 1]);
            accB.y += xp[(S + 2) >> 1].y;
            smA = div_const2(accA, Sf, yS);
            smB = div_const2(accB, Sf, yS);
            if (edge) {  // windows that leave the series: NaN (a NaN row in the sum, in the general kernel)
                const long t0 = pA.x;
                smA.x = (t0 - H >= 0 && t0 + H < T) ? smA.x : qnan;
                smA.y = (t0 + 1 - H >= 0 && t0 + 1 + H < T) ? smA.y : qnan;
                smB.x = (t0 + 2 - H >= 0 && t0 + 2 + H < T) ? smB.x : qnan;
                smB.y = (t0 + 3 - H >= 0 && t0 + 3 + H < T) ? smB.y : qnan;
            }
            // centre rows H .. H+3 of the four dayofyears
            const v2f xcA = (H & 1) ? (v2f){xp[(H - 1) / 2].y, xp[(H + 1) / 2].x} : xp[H / 2];
            const v2f xcB = (H & 1) ? (v2f){xp[(H + 1) / 2].y, xp[(H + 3) / 2].x} : xp[H / 2 + 1];
            const bool partial = pA.z < 4;  // only a prefix of the 4 dayofyears exists this year (leap day chunk)
            const bool has1 = pA.z > 1, has2 = pA.z > 2, has3 = pA.z > 3;
            if (!partial) {
                n_invalid += (finite_f(xcA.x) ? 0 : 1) + (finite_f(xcA.y) ? 0 : 1) + (finite_f(xcB.x) ? 0 : 1) +
                             (finite_f(xcB.y) ? 0 : 1);
            } else {
                n_invalid += finite_f(xcA.x) ? 0 : 1;
                if (has1) n_invalid += finite_f(xcA.y) ? 0 : 1; else smA.y = qnan;
                if (has2) n_invalid += finite_f(xcB.x) ? 0 : 1; else smB.x = qnan;
                smB.y = qnan;
            }
            if (pA.y >= 0) {  // output rows
                v2f sA = splat2(0.f), sB = splat2(0.f);
#pragma unroll
                for (int j = 0; j < ((SHIFT_EXP & 32) ? 2 : W); ++j) {
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
                const rsrc_t ro = make_rsrc(out + (size_t)pA.y * C);
                if (!(SHIFT_EXP & 4) || aA.x == 12345.678f) {
                    stb_f32(ro, voff, 0, write_clim ? climA.x : aA.x);
                    if (has1) stb_f32(ro, voff, rowb, write_clim ? climA.y : aA.y);
                    if (has2) stb_f32(ro, voff, 2 * rowb, write_clim ? climB.x : aB.x);
                    if (has3) stb_f32(ro, voff, 3 * rowb, write_clim ? climB.y : aB.y);
                }
                if ((do_bins || TAILS) && (!(SHIFT_EXP & 8) || aB.y == 12345.678f)) {
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
                        if (!TAILS) {  // the keys decide "counted or not" by comparing with the last edge instead
                            k0 = (a.x == a.x) ? k0 : nb;
                            k1 = (a.y == a.y) ? k1 : nb;
                        }
                    };
                    int k0, k1, k2, k3;
                    digit2(aA, k0, k1);
                    digit2(aB, k2, k3);
                    if (TAILS) {
                        // keys of this year's 4 samples; 0 for absent dayofyears and for samples the histogram drops: the
                        // bin is nb exactly when a >= edges[nb], and NaN fails the comparison too
                        const float l1 = has1 ? e_last : -__builtin_inff(), l2 = has2 ? e_last : -__builtin_inff(),
                                    l3 = has3 ? e_last : -__builtin_inff();  // uniform
                        const bool v0 = aA.x < e_last, v1 = aA.y < l1, v2 = aB.x < l2, v3 = aB.y < l3;
                        // ((k + 1) << 7) | pos = (k << 7) + (128 + pos), pos < 128 uniform
                        const unsigned b0 = 128u + (unsigned)(pA.w - t_ds0), b1 = 128u + (unsigned)(pB.x - t_ds1);
                        const unsigned b2 = 128u + (unsigned)(pB.y - t_ds2), b3 = 128u + (unsigned)(pB.z - t_ds3);
                        const unsigned q0 = v0 ? ((unsigned)k0 << TAIL_POS_BITS) + b0 : 0u, q1 = v1 ? ((unsigned)k1 << TAIL_POS_BITS) + b1 : 0u;
                        const unsigned q2 = v2 ? ((unsigned)k2 << TAIL_POS_BITS) + b2 : 0u, q3 = v3 ? ((unsigned)k3 << TAIL_POS_BITS) + b3 : 0u;
                        newkeys[wave][0][t_slot][lane] = q0 | (q1 << 16);
                        newkeys[wave][1][t_slot][lane] = q2 | (q3 << 16);
                        t_cntA += (v0 ? 1u : 0u) + (v1 ? 0x10000u : 0u);
                        t_cntB += (v2 ? 1u : 0u) + (v3 ? 0x10000u : 0u);
                        // not counted although it is a number: a value at or beyond the last edge (next to never: one test
                        // of the largest of the four, maxNum skips NaN)
                        const float mx = fmaxf(fmaxf(aA.x, has1 ? aA.y : aA.x), fmaxf(has2 ? aB.x : aA.x, has3 ? aB.y : aA.x));
                        if (__builtin_amdgcn_ballot_w64(mx >= e_last) != 0) {
                            t_ovfA |= ((aA.x >= e_last) ? 1u : 0u) | ((has1 && aA.y >= e_last) ? 0x10000u : 0u);
                            t_ovfB |= ((has2 && aB.x >= e_last) ? 1u : 0u) | ((has3 && aB.y >= e_last) ? 0x10000u : 0u);
                        }
                        ++t_slot;  // flushed between years (main loop), outside this body's register pressure
                    } else {
                        stb_u16(rbins, bin_lane, pA.w * 32, k0);
                        if (has1) stb_u16(rbins, bin_lane, pB.x * 32, k1);
                        if (has2) stb_u16(rbins, bin_lane, pB.y * 32, k2);
                        if (has3) stb_u16(rbins, bin_lane, pB.z * 32, k3);
                    }
                }
            }
        }
        rA[J + W] = smA;  // year y joins the history
        rB[J + W] = smB;
        SSTAMP(tsC);
        if (stage_next) stage_store((y + 1) & 1, nx);
        if (!(SHIFT_EXP & 1)) __syncthreads();
#ifdef SHIFT_STAMPS
        const unsigned long long tsD = __builtin_amdgcn_s_memtime();
        st_top += tsB - tsA;
        st_mid += tsC - tsB;
        st_end += tsD - tsC;
#endif
    };
    for (int y = 0; y < n_cal; y += 2) {
        one_year(y, std::integral_constant<int, 0>{});
        if (TAILS && t_slot == SHIFT_LIST) tails_flush();
        if (y + 1 < n_cal) one_year(y + 1, std::integral_constant<int, 1>{});
        if (TAILS && t_slot == SHIFT_LIST) tails_flush();
#pragma unroll
        for (int j = 0; j < W; ++j) {
            rA[j] = rA[j + 2];
            rB[j] = rB[j + 2];
        }
    }
    if (TAILS && mine) {
        if (t_slot > 0) tails_flush();
        while (t_list < tails.nper) tails_flush();  // dayofyears with fewer rows than the longest bucket: the remaining lists are empty
        // lists the walk never reached (fewer output years than lists * 16 cannot happen: nper = ceil(years / 16)) are not
        // read by anyone; the counts and flags of the 4 dayofyears:
        if (active) {
            const unsigned cn[4] = {t_cntA & 0xFFFFu, t_cntA >> 16, t_cntB & 0xFFFFu, t_cntB >> 16};
            const unsigned ov[4] = {t_ovfA & 0xFFFFu, t_ovfA >> 16, t_ovfB & 0xFFFFu, t_ovfB >> 16};
#pragma unroll
            for (int di = 0; di < 4; ++di)
                if (d0 + di < NDOY) tails.aux[(size_t)(d0 + di) * C + c] = (unsigned short)(cn[di] | (ov[di] ? 0x8000u : 0u));
        }
    }
    if (invalid_count && active && n_invalid) atomicAdd(&invalid_count[c], n_invalid);
#ifdef SHIFT_STAMPS
    if (TAILS && lane == 0 && tails.dbg) {
        atomicAdd(&tails.dbg[5], st_top);
        atomicAdd(&tails.dbg[6], st_mid);
        atomicAdd(&tails.dbg[7], st_end);
        atomicAdd(&tails.dbg[4], 1ull);
    }
#endif
}

struct ShiftArgs {
    const float* x;
    int64_t T, C;
    const int4* year_plan;
    const int4* fplan;
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
    TailOut tails;
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
                       a.mask, a.invalid_count, ncb, nchunks, MAREX_ABLATE_OPT(ctx, "SHIFT_ABLATE"), D == 4 ? a.skip : nullptr);
    HIP_TRY(ctx, hipGetLastError());
    return 0;
}

template <int W, int S = 21>
static void launch_shift_fast(marex_ctx* ctx, const ShiftArgs& a) {
    const int ncg = (int)((a.C + 63) / 64);
    if (a.tails.lists)
        hipLaunchKernelGGL((k_shift_fast<W, true, S>), dim3(xcd_grid(ncg, 23)), dim3(256), 0, ctx->stream, a.x, (long)a.T, (long)a.C,
                           a.fplan, a.n_cal, a.skip, a.write_clim, a.edges, a.nb, (long)a.T_out, a.out, a.bins, a.mask,
                           a.invalid_count, ncg, 23, a.tails);
    else
        hipLaunchKernelGGL((k_shift_fast<W, false, S>), dim3(xcd_grid(ncg, 23)), dim3(256), 0, ctx->stream, a.x, (long)a.T, (long)a.C,
                           a.fplan, a.n_cal, a.skip, a.write_clim, a.edges, a.nb, (long)a.T_out, a.out, a.bins, a.mask,
                           a.invalid_count, ncg, 23, a.tails);
}

template <int D, int WCAP, bool RREG>
static int dispatch_shifting_S(marex_ctx* ctx, const ShiftArgs& a) {
    if (a.S == 21) return launch_shifting<D, 21, true, WCAP, RREG>(ctx, a);
    return launch_shifting<D, 1, false, WCAP, RREG>(ctx, a);  // any other smoothing width: generic row loop
}

static int shifting_impl(marex_ctx* ctx, const char* who, const float* x, int64_t T, int64_t C, const int32_t* year_plan,
                         int n_cal_years, int W, int S, int write_clim, const float* edges, int nb, int64_t T_out, float* out,
                         uint16_t* bins, uint8_t* mask, int32_t* invalid_count, TailOut tails, bool* all_fast_possible) {
    if (!ctx) return -1;
    if (!x || !year_plan || !out || T <= 0 || C <= 0 || n_cal_years <= 0)
        return fail(ctx, -1, "%s: null pointer or empty shape", who);
    if (((uintptr_t)year_plan & 15) != 0) return fail(ctx, -1, "%s: year_plan must be 16-byte aligned", who);
    if (W < 1 || S < 1) return fail(ctx, -1, "%s: W and S must be >= 1", who);
    if (S > T) S = (int)T + 1;  // every window leaves the series: all-NaN smoothing either way
    if (W > 64) return fail(ctx, -4, "%s: window_year_baseline > 64 is not supported", who);
    if ((bins || tails.lists) && (!edges || nb < 4 || nb > 65534 || T_out <= 0))
        return fail(ctx, -1, "%s: binning needs edges, T_out and 4 <= nb <= 65534", who);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    ShiftArgs a{x, T, C, reinterpret_cast<const int4*>(year_plan), nullptr, n_cal_years, W, S, write_clim,
                edges, nb, T_out, out, bins, mask, invalid_count, nullptr, tails};
    // 4 dayofyears per workgroup (6 row loads per output) while the padded W-year LDS ring leaves room for two
    // workgroups per CU, otherwise one dayofyear
    const int forceD = ctx_opt(ctx, "SHIFT_D", 0);
    const int reg = ctx_opt(ctx, "SHIFT_RING", 1);  // 1 (default): history in registers, 0: LDS ring
    // regular chunks of the calendar go to k_shift_fast (S = 21, instantiated W, arange edge table)
    // instantiated (W, S): S = 21 with the W list, S = 11 / 15 with W = 5, 10, 15
    const bool fast_w = W == 3 || W == 4 || W == 5 || W == 6 || W == 7 || W == 10 || W == 13 || W == 15;
    const bool fast_ws = S == 21 ? fast_w : ((S == 11 || S == 15) && (W == 5 || W == 10 || W == 15));
    const bool fast_cfg = ctx_opt(ctx, "SHIFT_FAST", 1) != 0 && fast_ws && forceD == 0 && T >= S + 3 &&
                          T_out < (1 << 24) && C < (1 << 24) && MAREX_ABLATE_OPT(ctx, "SHIFT_ABLATE") == 0;
    if (all_fast_possible) *all_fast_possible = fast_cfg;
    if (fast_cfg && !ctx->shift_info) HIP_TRY(ctx, hipMalloc((void**)&ctx->shift_info, SHIFT_INFO_WORDS * sizeof(int)));
    if (fast_cfg && ctx->shift_plan_years < (size_t)n_cal_years) {
        if (ctx->shift_plan) {
            HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));  // an earlier launch may still read the old table
            (void)hipFree(ctx->shift_plan);
            ctx->shift_plan = nullptr;
            ctx->shift_plan_years = 0;
        }
        HIP_TRY(ctx, hipMalloc((void**)&ctx->shift_plan, (size_t)n_cal_years * 92 * 8 * sizeof(int)));
        ctx->shift_plan_years = (size_t)n_cal_years;
    }
    LaunchTimer lt(ctx, MAREX_K_SHIFTING);  // one timed region: classify + fast kernel + general kernel
    if (fast_cfg) {
        hipLaunchKernelGGL(k_shift_classify, dim3(1), dim3(1024), 0, ctx->stream, a.year_plan, n_cal_years, edges, nb,
                           (bins || tails.lists) ? 1 : 0, 1, ctx->shift_info, reinterpret_cast<int4*>(ctx->shift_plan));
        a.skip = ctx->shift_info;
        a.fplan = reinterpret_cast<const int4*>(ctx->shift_plan);
        if (S == 11 || S == 15) {
            switch (W * 100 + S) {
                case 511: launch_shift_fast<5, 11>(ctx, a); break;
                case 515: launch_shift_fast<5, 15>(ctx, a); break;
                case 1011: launch_shift_fast<10, 11>(ctx, a); break;
                case 1015: launch_shift_fast<10, 15>(ctx, a); break;
                case 1511: launch_shift_fast<15, 11>(ctx, a); break;
                default: launch_shift_fast<15, 15>(ctx, a); break;
            }
        } else
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

extern "C" int marex_shifting_baseline_f32(marex_ctx* ctx, const float* x, int64_t T, int64_t C,
                                           const int32_t* year_plan, int n_cal_years, int W, int S,
                                           int write_clim, const float* edges, int nb, int64_t T_out, float* out,
                                           uint16_t* bins, uint8_t* mask, int32_t* invalid_count) {
    return shifting_impl(ctx, "marex_shifting_baseline_f32", x, T, C, year_plan, n_cal_years, W, S, write_clim, edges, nb, T_out, out,
                         bins, mask, invalid_count, TailOut{nullptr, nullptr, nullptr, 0, nullptr}, nullptr);
}

// defined in marex_tails.hip: extraction restricted to the dayofyear chunks a flag table does NOT mark (skip == NULL: all)
int marex_tail_extract_impl(marex_ctx* ctx, const float* anom, int64_t T_out, int64_t C, const int32_t* doy_start,
                            const int32_t* doy_rows, int max_bucket, const float* edges, int nb, int list_rows, void* lists,
                            uint16_t* aux, const int* skip_chunks);

extern "C" int marex_shifting_baseline_tails_f32(marex_ctx* ctx, const float* x, int64_t T, int64_t C, const int32_t* year_plan,
                                                 int n_cal_years, int W, int S, const float* edges, int nb, int64_t T_out,
                                                 float* out, uint8_t* mask, int32_t* invalid_count, const int32_t* doy_start,
                                                 const int32_t* doy_rows, int max_bucket, void* lists, uint16_t* aux) {
    if (!ctx) return -1;
    if (!doy_start || !doy_rows || !lists || !aux) return fail(ctx, -1, "marex_shifting_baseline_tails_f32: null pointer");
    if (nb > TAIL_MAX_NB || max_bucket < 1 || max_bucket > 6 * SHIFT_LIST || C > (1 << 24))
        return fail(ctx, -4, "marex_shifting_baseline_tails_f32: shape outside the emitted tail format (nb <= %d, buckets <= %d rows)",
                    TAIL_MAX_NB, 6 * SHIFT_LIST);
    if (((uintptr_t)lists & 15) != 0) return fail(ctx, -1, "marex_shifting_baseline_tails_f32: lists must be 16-byte aligned");
    TailOut t{reinterpret_cast<uint4*>(lists), aux, doy_start, (max_bucket + SHIFT_LIST - 1) / SHIFT_LIST, ctx_debug_counters(ctx)};
    bool fast = false;
    const int rc = shifting_impl(ctx, "marex_shifting_baseline_tails_f32", x, T, C, year_plan, n_cal_years, W, S, 0, edges, nb, T_out,
                                 out, nullptr, mask, invalid_count, t, &fast);
    if (rc != 0) return rc;
    // dayofyear chunks the fast kernel did not take (irregular calendars, other S / W): their tails from the anomalies
    return marex_tail_extract_impl(ctx, out, T_out, C, doy_start, doy_rows, max_bucket, edges, nb, SHIFT_LIST, lists, aux,
                                   fast ? ctx->shift_info : nullptr);
}
