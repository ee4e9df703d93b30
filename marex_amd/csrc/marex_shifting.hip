// marex_shifting.hip -- K_A: validation + smoothing + rolling climatology + anomaly + bins
#include "marex_common.hip.h"
#include "marex_tails.hip.h"
#include <numeric>
#include <utility>

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
    asm volatile("buffer_store_dwordx4 %0, %1, %2, %3 offen nt\n\ts_nop 2" : : "v"(v), "v"(voff), "s"(r), "s"(soff_u) : "memory");
}


// LDS-DMA staging of the year's rows (`buffer_load_dword[x4] ... offen lds`: memory -> LDS with no VGPR destination, no
// ds_write; addressing pinned by profiles/tools/lds_dma_probe.hip: LDS byte = M0 + lane * size, memory byte = descriptor base
// + soffset + per-lane voffset).  One dwordx4 instruction moves FOUR rows of the stage (lanes 16 r .. 16 r + 15 take row r:
// 16 lanes x 16 bytes = the 64 cells of a row), a dword instruction one row.  hipcc neither counts these loads nor knows
// that M0 is written: every statement saves / restores M0, and the kernel waits for them with its own s_waitcnt vmcnt(N).
// The same loads pointed at a scratch kilobyte of LDS are L2 PREFETCHES (nothing ever reads the bytes): the rows a workgroup
// will stage a year from now are pulled out of HBM a year early, so the staging loads themselves are L2 hits.
template <int RPW>
struct DmaRows;  // G4 dwordx4 + G1 dword instructions cover the RPW rows of a wave
#define MAREX_DMA_X4(m, so) "s_mov_b32 m0, " m "\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %3, " so " offen lds\n\t"
#define MAREX_DMA_X1(m, so) "s_mov_b32 m0, " m "\n\ts_nop 0\n\tbuffer_load_dword %2, %3, " so " offen lds\n\t"
template <>
struct DmaRows<9> {  // S = 21: rows 0-3, 4-7, 8
    static constexpr int NL = 3;
    static __device__ __forceinline__ void issue(tl_out_rsrc_t r, unsigned voff4, unsigned voff1, unsigned l0, unsigned lstep, unsigned so0, unsigned rowb) {
        unsigned keep;
        const unsigned l1 = l0 + 4 * lstep, l2 = l0 + 8 * lstep, so1 = so0 + 4 * rowb, so2 = so0 + 8 * rowb;
        asm volatile("s_mov_b32 %0, m0\n\t" MAREX_DMA_X4("%4", "%7") MAREX_DMA_X4("%5", "%8") MAREX_DMA_X1("%6", "%9") "s_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(voff4), "v"(voff1), "s"(r), "s"(l0), "s"(l1), "s"(l2), "s"(so0), "s"(so1), "s"(so2) : "memory");
    }
};
template <>
struct DmaRows<8> {  // S = 15 (30 rows staged as 32): rows 0-3, 4-7
    static constexpr int NL = 2;
    static __device__ __forceinline__ void issue(tl_out_rsrc_t r, unsigned voff4, unsigned voff1, unsigned l0, unsigned lstep, unsigned so0, unsigned rowb) {
        unsigned keep;
        const unsigned l1 = l0 + 4 * lstep, so1 = so0 + 4 * rowb;
        asm volatile("s_mov_b32 %0, m0\n\t" MAREX_DMA_X4("%4", "%6") MAREX_DMA_X4("%5", "%7") "s_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(voff4), "v"(voff1), "s"(r), "s"(l0), "s"(l1), "s"(so0), "s"(so1) : "memory");
    }
};
template <>
struct DmaRows<7> {  // S = 11 (26 rows staged as 28): rows 0-3, 4, 5, 6
    static constexpr int NL = 4;
    static __device__ __forceinline__ void issue(tl_out_rsrc_t r, unsigned voff4, unsigned voff1, unsigned l0, unsigned lstep, unsigned so0, unsigned rowb) {
        unsigned keep;
        const unsigned l1 = l0 + 4 * lstep, l2 = l0 + 5 * lstep, l3 = l0 + 6 * lstep;
        const unsigned so1 = so0 + 4 * rowb, so2 = so0 + 5 * rowb, so3 = so0 + 6 * rowb;
        asm volatile("s_mov_b32 %0, m0\n\t" MAREX_DMA_X4("%4", "%8") MAREX_DMA_X1("%5", "%9") MAREX_DMA_X1("%6", "%10") MAREX_DMA_X1("%7", "%11") "s_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(voff4), "v"(voff1), "s"(r), "s"(l0), "s"(l1), "s"(l2), "s"(l3), "s"(so0), "s"(so1), "s"(so2), "s"(so3) : "memory");
    }
};
#undef MAREX_DMA_X4
#undef MAREX_DMA_X1

#define SHIFT_INFO_WORDS 128  // [0..91] chunk handled by the fast kernel, [92] arange edges usable
#define CLASSIFY_SPLIT 11      // threads per chunk in k_shift_classify (92 * 11 = 1012 <= 1024)
#define LEAN_CHUNKS 96         // chunk slots per year of lean records (92 chunks, padded to whole workgroups of 8 waves)

// fplan[year][chunk] = two int4: {timestep of the chunk's first dayofyear (-1: absent), its output row (-1: none), number of
// leading dayofyears present (0..4), bin-matrix row of dayofyear 0} and {bin-matrix rows of dayofyears 1..3, timestep of the
// first dayofyear of the chunk's WORKGROUP (16 dayofyears) in the FOLLOWING year}: everything k_shift_fast needs per year
// in ONE 32-byte scalar load, every field live (a dead field lets the register allocator recycle its SGPR and wait for
// the load on the spot)
// lplan != NULL (k_shift_lean, smoothing width reg_S): the lean records of that kernel (LeanRec, below) -- a year is REGULAR for a
// chunk when all 4 dayofyears are present on consecutive timesteps, the workgroup's rows are staged (block inside the series,
// the chunk where the block puts it), no smoothing window leaves the series and, in output years, the 4 samples sit at the same
// position of their buckets.
__global__ void k_shift_classify(const int4* __restrict__ year_plan, int n_cal, const float* __restrict__ edges, int nb,
                                 int want_bins, int enable, int* __restrict__ info, int4* __restrict__ fplan, long T, long C,
                                 int reg_S, int4* __restrict__ lplan, const int* __restrict__ doy_start, int wpb, int lean_bins) {
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
    if (chunk < 92 && lplan && reg_S > 0) {  // lean records of years -1 .. n_cal
        // wpb = waves (chunks of 4 dayofyears) per workgroup of the lean kernel: 4 or 8
        const int H = reg_S / 2, NPAIR = (reg_S + 3) / 2, NROWS = wpb * ((reg_S + 4 * wpb - 1 + wpb - 1) / wpb);
        const int d0 = chunk * 4, blk16 = (chunk / wpb) * wpb * 4;
        // bin-matrix output (lean_bins): a regular year's 4 bin rows are row0 + {0, D1, D2, D3} with the SAME D every year (the
        // distance of the dayofyears' buckets); D comes from the first year whose 4 dayofyears all have a bin row and travels in
        // the padding record of "year -1"
        int D1 = 0, D2 = 0, D3 = 0;
        if (lean_bins && d0 + 3 < NDOY)
            for (int y2 = 0; y2 < n_cal; ++y2) {
                const int z0 = year_plan[(size_t)y2 * NDOY + d0].z, z1 = year_plan[(size_t)y2 * NDOY + d0 + 1].z;
                const int z2 = year_plan[(size_t)y2 * NDOY + d0 + 2].z, z3 = year_plan[(size_t)y2 * NDOY + d0 + 3].z;
                if (z0 >= 0 && z1 >= 0 && z2 >= 0 && z3 >= 0) {
                    D1 = z1 - z0;
                    D2 = z2 - z0;
                    D3 = z3 - z0;
                    break;
                }
            }
        for (int yy = sub - 1; yy <= n_cal; yy += CLASSIFY_SPLIT) {
            int flags = 0, tbv = -1, pos = 0;
            long long xoff = 0, ooff = 0;
            if (yy + 1 >= 0 && yy + 1 < n_cal) {  // next year's rows of the workgroup: rows tb - H .. tb - H + NROWS - 1
                const int tbn = year_plan[(size_t)(yy + 1) * NDOY + blk16].x;
                if (tbn >= H && (long)tbn - H + NROWS <= T) {
                    flags |= 4;  // LR_STAGE_NEXT
                    xoff = (long long)(tbn - H) * C * 4;
                }
            }
            if (yy >= 0 && yy < n_cal) {
                int4 e[4];
                for (int i = 0; i < 4; ++i) e[i] = (d0 + i < NDOY) ? year_plan[(size_t)yy * NDOY + d0 + i] : make_int4(-1, -1, -1, 0);
                tbv = year_plan[(size_t)yy * NDOY + blk16].x;
                int reg = d0 + 3 < NDOY && e[0].x >= 0 && e[1].x >= 0 && e[2].x >= 0 && e[3].x >= 0;
                reg = reg && tbv >= H && (long)tbv - H + NROWS <= T && e[0].x == tbv + 4 * (chunk % wpb);
                reg = reg && e[0].x - H >= 0 && (long)e[0].x - H + 2 * NPAIR <= T;
                for (int i = 1; i < 4; ++i) reg = reg && e[i].x == e[0].x + i && ((e[0].y >= 0) == (e[i].y >= 0));
                if (e[0].y >= 0) {
                    flags |= 2;  // LR_OUT
                    ooff = (long long)e[0].y * C * 4;
                    if (lean_bins) {
                        pos = e[0].z;  // the first bin row itself
                        reg = reg && pos >= 0 && e[1].z == pos + D1 && e[2].z == pos + D2 && e[3].z == pos + D3;
                        for (int i = 1; i < 4; ++i) reg = reg && e[i].y == e[0].y + i;
                    } else {
                        pos = e[0].z - doy_start[d0];
                        reg = reg && pos >= 0 && pos < TAIL_MAX_BUCKET;
                        for (int i = 1; i < 4; ++i) reg = reg && e[i].y == e[0].y + i && e[i].z - doy_start[d0 + i] == pos;
                    }
                }
                if (reg) flags |= 1;  // LR_REG
            }
            int4* r = lplan + ((size_t)(yy + 1) * LEAN_CHUNKS + chunk) * 2;
            r[0] = make_int4(flags, tbv, (int)(unsigned)(xoff & 0xFFFFFFFFll), (int)(unsigned)((unsigned long long)xoff >> 32));
            r[1] = make_int4((int)(unsigned)(ooff & 0xFFFFFFFFll), (int)(unsigned)((unsigned long long)ooff >> 32), pos, 0);
            if (yy < 0 && lean_bins) r[1] = make_int4(0, D1, D2, D3);
            // the waves of the last workgroup that have no dayofyears (chunks 92..95 with 8 waves) only help staging: the
            // workgroup's part of the record, nothing of a chunk
            if (chunk + 4 >= 92 && chunk + 4 < (92 + wpb - 1) / wpb * wpb) {
                r[8] = make_int4(flags & 4, tbv, (int)(unsigned)(xoff & 0xFFFFFFFFll), (int)(unsigned)((unsigned long long)xoff >> 32));
                r[9] = make_int4(0, 0, 0, 0);
            }
        }
    }
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
    unsigned* aux;             // [366][C] count | samples beyond the table (marex_tails.hip.h)
    const int* doy_start;      // [367] first bin-matrix row of every dayofyear (key positions = row - doy_start)
    int nper;
    unsigned long long* dbg;   // debug counters (-DSHIFT_STAMPS builds: phase timers)
};

template <int W, bool TAILS, int S = 21, bool DMA = false>
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
    // DMA (TAILS only): the rows go from memory straight into the stage (DmaRows above) and the rows of the year AFTER next are
    // prefetched into the L2 through a scratch kilobyte; every wave issues the same instructions, so the stage has 4 RPW rows
    static_assert(!DMA || TAILS, "k_shift_fast: LDS-DMA staging is built for the single stage buffer of the TAILS variant");
    constexpr int NROWS = DMA ? 4 * RPW : NST;
    __shared__ float stage[NSTAGE][NROWS * 64];
    __shared__ float pf_dump[DMA ? 256 : 1];
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
    auto stage_ok = [&](int tb) { return tb >= H && (long)tb - H + NROWS <= T; };
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
    // DMA: lane offsets of the 4-row instruction (lane 16 r + s: row r, cells 4 s .. 4 s + 3 of the group; a segment beyond C --
    // C is a multiple of 4 on this path -- reads the last four cells instead: those lanes are masked wherever they store)
    typedef __attribute__((address_space(3))) float lds_f32;
    unsigned dma_voff4 = 0, dma_lds = 0, dma_dump = 0;
    if (DMA) {
        const long c4 = (long)cg * 64 + (lane & 15) * 4;
        dma_voff4 = (unsigned)(lane >> 4) * (unsigned)rowb + (unsigned)(c4 + 4 <= C ? c4 : C - 4) * 4u;
        dma_lds = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(size_t)(lds_f32*)&stage[0][(RPW * wave) * 64]);
        dma_dump = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(size_t)(lds_f32*)&pf_dump[0]);
    }
    // rows 365 days after a staged block lie inside the series (the prefetch of the year after next: 365 or 366 days on, the
    // difference is one row of 4 RPW)
    auto pf_ok = [&](int tb) { return (long)tb + 365 - H + NROWS <= T; };
    // stage the rows of the year whose workgroup block starts at timestep tb (+ prefetch the same rows 365 days on)
    auto dma_year = [&](int tb, bool pf) {
        const tl_out_rsrc_t rs = tl_out_make_rsrc(x + (size_t)(tb - H + RPW * wave) * C);
        DmaRows<RPW>::issue(rs, dma_voff4, voff, dma_lds, 256u, 0u, (unsigned)rowb);
        if (pf) DmaRows<RPW>::issue(rs, dma_voff4, voff, dma_dump, 0u, 365u * (unsigned)rowb, (unsigned)rowb);
    };
    bool dma_pf = false;
    if (DMA) {
        if (stage_ok(tb_cur)) {
            dma_year(tb_cur, pf_ok(tb_cur));
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        asm volatile("s_barrier" ::: "memory");
    } else {
        float nx[RPW];
        if (stage_ok(tb_cur)) {
            stage_load(tb_cur, nx);
            stage_store(0, nx);
        }
        __syncthreads();
    }

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
                        if (!DMA || active) tl_out_store(rl, lvoff, (unsigned)li * (unsigned)C * 16u, w);
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
        float nx[RPW];  // (unused with DMA)
        // next year's rows, one iteration ahead (its first timestep arrived during the previous iteration)
        const bool stage_next = y + 1 < n_cal && stage_ok(tb_n1) && !(SHIFT_EXP & 2);
        if (!DMA && stage_next) stage_load(tb_n1, nx);
        v2f smA = splat2(qnan), smB = splat2(qnan);
        v2f oA = splat2(0.f), oB = splat2(0.f);  // DMA: this year's output rows, stored at the end
        int o_n = 0;
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
            if (DMA && stage_next) {
                // the stage is free: next year's rows on their way into it, the rows of the year after next on their way into the L2
                dma_pf = y + 2 < n_cal && pf_ok(tb_n1);
                dma_year(tb_n1, dma_pf);
            }
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
                if (DMA) {  // stored at the end of the year, behind the wait for the staged rows (the stores stay younger than it)
                    oA = write_clim ? climA : aA;
                    oB = write_clim ? climB : aB;
                    o_n = pA.z;
                } else {
                const rsrc_t ro = make_rsrc(out + (size_t)pA.y * C);
                if (!(SHIFT_EXP & 4) || aA.x == 12345.678f) {
                    stb_f32(ro, voff, 0, write_clim ? climA.x : aA.x);
                    if (has1) stb_f32(ro, voff, rowb, write_clim ? climA.y : aA.y);
                    if (has2) stb_f32(ro, voff, 2 * rowb, write_clim ? climB.x : aB.x);
                    if (has3) stb_f32(ro, voff, 3 * rowb, write_clim ? climB.y : aB.y);
                }
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
                        if (__builtin_amdgcn_ballot_w64(mx >= e_last) != 0) {  // their positions (the first two) go into the aux word
                            unsigned s0 = t_ovfA & 0xFFFFu, s1 = t_ovfA >> 16, s2 = t_ovfB & 0xFFFFu, s3 = t_ovfB >> 16;
                            if (aA.x >= e_last) s0 = tail_ovf_add(s0, (unsigned)(pA.w - t_ds0));
                            if (has1 && aA.y >= e_last) s1 = tail_ovf_add(s1, (unsigned)(pB.x - t_ds1));
                            if (has2 && aB.x >= e_last) s2 = tail_ovf_add(s2, (unsigned)(pB.y - t_ds2));
                            if (has3 && aB.y >= e_last) s3 = tail_ovf_add(s3, (unsigned)(pB.z - t_ds3));
                            t_ovfA = s0 | (s1 << 16);
                            t_ovfB = s2 | (s3 << 16);
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
        if (DMA) {
            // the staged rows of next year have landed (everything older than the prefetch loads issued behind them), then the
            // anomaly stores, then the barrier that lets every wave read the stage
            if (stage_next) {
                if (dma_pf)
                    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DmaRows<RPW>::NL) : "memory");
                else
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            if (o_n > 0 && active) {  // lanes beyond C hold other cells' rows on this path: they store nothing
                const rsrc_t ro = make_rsrc(out + (size_t)pA.y * C);
                stb_f32(ro, voff, 0, oA.x);
                if (o_n > 1) stb_f32(ro, voff, rowb, oA.y);
                if (o_n > 2) stb_f32(ro, voff, 2 * rowb, oB.x);
                if (o_n > 3) stb_f32(ro, voff, 3 * rowb, oB.y);
            }
            asm volatile("s_barrier" ::: "memory");
        } else {
        if (stage_next) stage_store((y + 1) & 1, nx);
        if (!(SHIFT_EXP & 1)) __syncthreads();
        }
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
                if (d0 + di < NDOY) tails.aux[(size_t)(d0 + di) * C + c] = tail_aux_word(cn[di], ov[di]);
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

// ------------------------------------------------------------------------------------------------
// K_A lean: k_shift_fast<W, TAILS> rewritten around its instruction count.  The fast kernel runs at the issue limit of the
// SIMDs (223 vector + 110 scalar + 22 branch instructions per wave, year and 4 dayofyears; staging the rows by LDS-DMA with an
// L2 prefetch a year ahead changed nothing), so this kernel removes instructions:
//   * REGULAR years -- flagged per (year, chunk) by k_shift_classify: all 4 dayofyears present, rows staged, no window leaving
//     the series, the 4 samples at the same bucket position -- take a straight-line body that never looks at the calendar;
//     every other year (first days of the series, the leap-day chunk, the trailing partial year) takes the general body of
//     k_shift_fast;
//   * (tried and dropped: a CIRCULAR history line, a run of regular years unrolled W times so that no register ever moves -- the
//     26 v_mov per year of the shift line gone, but 15 copies of the year body, with their rare blocks and the list flush inline,
//     are 180 KB of code and spill: 45.6 ms against 11.8 ms on a 100-yr band.  The line of W + 2 entries that moves by two
//     every two years stays.)
//   * everything about the calendar is resolved ONCE, on the device, into a lean record per (year, chunk) (flags, the byte
//     offsets of next year's staged rows and of this year's output rows, the bucket position): the straight-line body adds two
//     64-bit offsets and tests three flag bits where the fast kernel spends ~110 scalar instructions per year on records,
//     range checks and 64-bit address arithmetic;
//   * rows are staged by LDS-DMA (3 instructions per wave and year instead of 9 loads + 9 ds_write) and read from the stage in
//     two halves through the same registers.
//   (Also tried: lane masks of the class / range tests compared with ONE land mask by scalar instructions, scalar counters for
//   the uniform part -- 30 vector instructions fewer, 31 scalar instructions more per year, same time.)
// Same arithmetic, same bits: tests run both kernels against the oracle and against each other (SHIFT_LEAN=0 selects the fast one).
// ------------------------------------------------------------------------------------------------
template <typename F, int... Is>
__device__ __forceinline__ void unroll_steps_impl(F& f, std::integer_sequence<int, Is...>) {
    (f(std::integral_constant<int, Is>{}), ...);
}
template <int N, typename F>
__device__ __forceinline__ void unroll_steps(F&& f) {
    unroll_steps_impl(f, std::make_integer_sequence<int, N>{});
}

// lean records (k_shift_classify with reg_S > 0): lplan[((y + 1) * 92 + chunk) * 2 + {0, 1}], y = -1 .. n_cal (the first and the
// last year are padding: the prologue reads "year -1" for the staging of year 0, the prefetch of the last year's successor needs
// no range check).  A = {flags, first timestep of the workgroup's 16 dayofyears this year (-1: none), byte offset lo / hi of row
// (that timestep of NEXT year - H) in x}, B = {byte offset lo / hi of this year's first output row of the chunk, bucket position
// of its 4 samples, 0}.  Everything the straight-line body needs to know about the calendar is here, resolved on the device once.
#define LR_REG 1         // regular year for this chunk (see k_shift_classify)
#define LR_OUT 2         // an output year
#define LR_STAGE_NEXT 4  // next year's rows of the workgroup can be staged (the block lies inside the series)

// BINS: the 2-byte bin matrix of k_shift_fast<W, false> (short series: the band threshold kernel reads it) instead of the key lists
template <int W, int S, int NWV, bool BINS>  // NWV waves = 4 NWV dayofyears per workgroup: 4 (36 rows staged for 16) or 8 (52 for 32)
__global__ void __launch_bounds__(64 * NWV, 16 / NWV)  // sixteen waves per CU: at most 128 VGPRs
k_shift_lean(const float* __restrict__ x, int T, int C, const int4* __restrict__ fplan, const int4* __restrict__ lplan, int n_cal,
             const int* __restrict__ info, const float* __restrict__ edges, int nb, float* __restrict__ out,
             unsigned char* __restrict__ mask, int* __restrict__ invalid_count, int ncg, int nblk, TailOut tails,
             unsigned short* __restrict__ bins, long T_out) {
    int cg, bc;
    if (!xcd_swizzle(blockIdx.x, ncg, nblk, cg, bc)) return;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int lane = (int)(threadIdx.x & 63);
    const int chunk = bc * NWV + wave;
    static_assert((S & 1) == 1 && S >= 5 && S <= 25, "k_shift_lean: odd smoothing widths 5..25");
    static_assert(NWV == 4 || NWV == 8, "k_shift_lean: 4 or 8 waves");
    constexpr int H = S / 2, NPAIR = (S + 3) / 2, NST = S + 4 * NWV - 1, RPW = (NST + NWV - 1) / NWV;
    // the rows of a year go from memory straight into the stage by LDS-DMA (DmaRows above: three instructions per wave instead of
    // nine loads and nine ds_write, and no registers held by a prefetched year); every wave issues the same instructions, so
    // the stage has NWV RPW rows
    constexpr int NROWS = NWV * RPW;
    __shared__ float stage[NROWS * 64];
    __shared__ unsigned newkeys[BINS ? 1 : NWV][2][BINS ? 1 : SHIFT_LIST][64];  // [wave][pair of dayofyears][year slot][lane]
    const bool mine = chunk < 92 && info[chunk] != 0;  // wave-uniform; the other waves only help staging
    const int d0 = mine ? chunk * 4 : 0;
    const int c = cg * 64 + lane;
    const bool active = c < C;
    const unsigned cidx = active ? (unsigned)c : (unsigned)(C - 1);
    const unsigned voff = cidx * 4u;
    const int rowb = C * 4;
    const float e_first = edges[1], e_delta = edges[2] - edges[1], e_last = edges[nb];
    const float inv_width = (float)(nb - 1) / (e_last - e_first);
    constexpr float Sf = (float)S;
    const float yS = 1.0f / Sf;
    const float Wf = (float)W;
    const float yW = 1.0f / Wf;
    const float nbm1f = (float)(nb - 1);
    const float c0 = (1.0f - e_first * inv_width) - 0.0078125f;
    const float qnan = nan_f();
    if (chunk == 0 && mine && mask && active) mask[c] = finite_f(x[c]) ? 1 : 0;
    // bin matrix (BINS): lane part of the element index relative to the wave's first 16-cell block, as in k_shift_fast
    const unsigned bin_lane = (((cidx >> 4) - (unsigned)(cg * 4)) * (unsigned)T_out * 16u + (cidx & 15u)) * 2u;  // bytes
    const rsrc_t rbins = make_rsrc(BINS ? bins + (size_t)(cg * 4) * (size_t)T_out * 16 : nullptr);
    int bD1 = 0, bD2 = 0, bD3 = 0;  // bin rows of dayofyears 1..3 relative to dayofyear 0 (regular years)

    // lane offsets of the 4-row instruction (lane 16 r + s: row r, cells 4 s .. 4 s + 3 of the group; a segment beyond C -- C is a
    // multiple of 4 here -- reads the last four cells instead: those lanes are masked wherever they store)
    typedef __attribute__((address_space(3))) float lds_f32;
    const int c4 = cg * 64 + (lane & 15) * 4;
    const unsigned dma_voff4 = (unsigned)(lane >> 4) * (unsigned)rowb + (unsigned)(c4 + 4 <= C ? c4 : C - 4) * 4u;
    const unsigned dma_lds = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(size_t)(lds_f32*)&stage[(RPW * wave) * 64]);
    const char* xw = reinterpret_cast<const char*>(x) + (size_t)(RPW * wave) * (size_t)C * 4;  // this wave's share of a staged block
    auto dma_rows = [&](int lo, int hi) __attribute__((always_inline)) {
        const unsigned long long off = ((unsigned long long)(unsigned)hi << 32) | (unsigned)lo;
        DmaRows<RPW>::issue(tl_out_make_rsrc(xw + off), dma_voff4, voff, dma_lds, 256u, 0u, (unsigned)rowb);
    };
    constexpr int LSTRIDE = LEAN_CHUNKS * 2;  // int4 per year of lean records
    const int4* lr_next = lplan + (size_t)chunk * 2;  // "year -1"
    {
        const int4 pre = lr_next[0];
        if (BINS) {
            const int4 preB = lr_next[1];
            bD1 = preB.y;
            bD2 = preB.z;
            bD3 = preB.w;
        }
        if (pre.x & LR_STAGE_NEXT) {
            dma_rows(pre.z, pre.w);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        asm volatile("s_barrier" ::: "memory");
        lr_next += LSTRIDE;
    }
    int4 nA = lr_next[0], nB = lr_next[1];  // year 0
    lr_next += LSTRIDE;

    // history of dayofyears (0,1) and (2,3) as register lines; the year loop is unrolled by U: year J of a group reads entries
    // [J, J + W) and appends at [J + W], then the line moves down by U -- 4 W register moves per U years
    constexpr int U = W >= 13 ? 3 : 2;  // W = 15: 122 VGPRs with three years per group, spills with four
    v2f hA[W + U], hB[W + U];
#pragma unroll
    for (int j = 0; j < W + U; ++j) hA[j] = hB[j] = splat2(qnan);
    int n_invalid = 0;
    unsigned t_cntA = 0, t_cntB = 0, t_ovfA = 0, t_ovfB = 0;
    int t_slot = 0, t_list = 0;
    int n_lean = 0, n_gen = 0;  // years of this wave that took the straight-line / the general body (debug counters 6 / 7)
    auto tails_flush = [&]() __attribute__((always_inline)) {
        const tl_out_rsrc_t rl = tl_out_make_rsrc(tails.lists + (size_t)d0 * tails.nper * 2 * (size_t)C);
        const unsigned lvoff = cidx * 16u;
#pragma unroll
        for (int pr = 0; pr < 2; ++pr) {
            unsigned v[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) v[u] = (u < SHIFT_LIST && u < t_slot) ? newkeys[wave][pr][u][lane] : 0u;  // uniform bound
            sort16_desc(v);
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int di = 2 * pr + h;
                if (d0 + di < NDOY) {
#pragma unroll
                    for (int jj = 0; jj < 2; ++jj) {
                        unsigned w[4];
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const unsigned a = v[8 * jj + 2 * i], b = v[8 * jj + 2 * i + 1];
                            w[i] = h == 0 ? ((a & 0xFFFFu) | (b << 16)) : ((a >> 16) | (b & 0xFFFF0000u));
                        }
                        const int li = __builtin_amdgcn_readfirstlane((di * tails.nper + t_list) * 2 + jj);
                        if (active) tl_out_store(rl, lvoff, (unsigned)li * (unsigned)C * 16u, w);
                    }
                }
            }
        }
        ++t_list;
        t_slot = 0;
    };

    // mid: once per wave and year, behind the wave's last read of the stage -- the barrier that frees the stage, next year's rows
    // on their way into it, the record of the year after asked for
    auto mid = [&](const int4& cA) __attribute__((always_inline)) {
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        if (cA.x & LR_STAGE_NEXT) dma_rows(cA.z, cA.w);
        nA = lr_next[0];
        nB = lr_next[1];
        lr_next += LSTRIDE;
    };
    // Rows r0 .. r0 + 2 NPAIR - 1 (r0 = first timestep - H) come in TWO halves through the same registers (the 24 rows of a
    // year at once are 12 registers this kernel does not have): pairs [0, NH0) then [NH0, NPAIR).  how: 0 the stage, 1 memory,
    // 2 memory with rows outside the series clamped (the caller replaces the affected sums by NaN).
    constexpr int NH0 = (NPAIR + 1) / 2, NH1 = NPAIR - NH0;
    static_assert(NH0 >= 2, "the second chain starts in the first half");
    v2f xh[NH0];
    v2f accA = splat2(0.f), accB = splat2(0.f);
    float xc0 = 0.f, xc1 = 0.f, xc2 = 0.f, xc3 = 0.f;
    auto load_half = [&](auto Pc, int how, int ts) __attribute__((always_inline)) {
        constexpr int P = decltype(Pc)::value, M0 = P ? NH0 : 0, MN = P ? NH1 : NH0;
        if (how == 0) {
            const float* st = &stage[(4 * wave) * 64 + lane];
#pragma unroll
            for (int m = 0; m < MN; ++m) {
                xh[m].x = st[(2 * (M0 + m)) * 64];
                xh[m].y = st[(2 * (M0 + m) + 1) * 64];
            }
        } else if (how == 1) {
            const rsrc_t rx = make_rsrc(x + (size_t)(ts - H) * (size_t)C);
#pragma unroll
            for (int m = 0; m < MN; ++m) {
                xh[m].x = ldb_f32(rx, voff, (2 * (M0 + m)) * rowb);
                xh[m].y = ldb_f32(rx, voff, (2 * (M0 + m) + 1) * rowb);
            }
        } else {
#pragma unroll
            for (int m = 0; m < MN; ++m) {
                int ra = ts - H + 2 * (M0 + m), rb = ra + 1;
                ra = ra < 0 ? 0 : (ra > T - 1 ? T - 1 : ra);
                rb = rb < 0 ? 0 : (rb > T - 1 ? T - 1 : rb);
                xh[m].x = ldb_f32(make_rsrc(x + (size_t)ra * (size_t)C), voff, 0);
                xh[m].y = ldb_f32(make_rsrc(x + (size_t)rb * (size_t)C), voff, 0);
            }
        }
    };
    // sequential sums of rows i .. i+S-1 for the four dayofyears i = 0..3 (two chains per instruction, skewed by one row) and
    // the centre rows H .. H+3, the part that the pairs of half P feed
    auto half = [&](auto Pc) __attribute__((always_inline)) {
        constexpr int P = decltype(Pc)::value, M0 = P ? NH0 : 0, M1 = P ? NPAIR : NH0;
        if constexpr (S == 21) {
            // The two chains of a half as ONE block: a v_pk_add_f32 in its own asm statement (and, through an encoding alias of
            // the op_sel_hi bit of src0 with VOP3's dst op_sel, every compiler-visible v_pk_add_f32 too) is taken for a producer
            // of a partial register write, and the hazard recogniser puts an s_nop in front of its consumer -- 20 per year.
            // v_pk_add_f32 writes two whole registers: there is no such hazard, and inside a block nothing is inserted.
#define PK_LO(acc, src) "v_pk_add_f32 %" #acc ", %" #acc ", %" #src " op_sel_hi:[1,0]\n\t"
#define PK_HI(acc, src) "v_pk_add_f32 %" #acc ", %" #acc ", %" #src " op_sel:[0,1] op_sel_hi:[1,1]\n\t"
#define PK_BOTH(src) PK_LO(0, src) PK_LO(1, src) PK_HI(0, src) PK_HI(1, src)
            if (P == 0) {  // rows 0..11: chain A takes rows 1..11, chain B (starts at row 2) rows 3..11
                accA = (v2f){xh[0].x, -0.0f};
                accB = (v2f){xh[1].x, -0.0f};
                asm(PK_HI(0, 2) PK_LO(0, 3) PK_HI(0, 3) PK_HI(1, 3) PK_BOTH(4) PK_BOTH(5) PK_BOTH(6) PK_BOTH(7)
                    : "+v"(accA), "+v"(accB)
                    : "v"(xh[0]), "v"(xh[1]), "v"(xh[2]), "v"(xh[3]), "v"(xh[4]), "v"(xh[5]));
                xc0 = xh[5].x;  // centre rows H .. H+3 = 10 .. 13
                xc1 = xh[5].y;
            } else {  // rows 12..23: chain A rows 12..20 and its second lane row 21, chain B rows 12..22 and row 23
                asm(PK_BOTH(2) PK_BOTH(3) PK_BOTH(4) PK_BOTH(5) PK_LO(0, 6) PK_LO(1, 6) PK_HI(1, 6) PK_LO(1, 7)
                    : "+v"(accA), "+v"(accB)
                    : "v"(xh[0]), "v"(xh[1]), "v"(xh[2]), "v"(xh[3]), "v"(xh[4]), "v"(xh[5]));
                accA.y += xh[4].y;
                accB.y += xh[5].y;
                xc2 = xh[0].x;
                xc3 = xh[0].y;
            }
#undef PK_BOTH
#undef PK_HI
#undef PK_LO
            return;
        }
        if (P == 0) accA = (v2f){xh[0].x, -0.0f};
#pragma unroll
        for (int s = 1; s <= S - 1; ++s)
            if ((s >> 1) >= M0 && (s >> 1) < M1) accA = (s & 1) ? pk_add_bc_hi(accA, xh[(s >> 1) - M0]) : pk_add_bc_lo(accA, xh[(s >> 1) - M0]);
        if ((S >> 1) >= M0 && (S >> 1) < M1) accA.y += xh[(S >> 1) - M0].y;
        if (1 >= M0 && 1 < M1) accB = (v2f){xh[1 - M0].x, -0.0f};
#pragma unroll
        for (int s = 3; s <= S + 1; ++s)
            if ((s >> 1) >= M0 && (s >> 1) < M1) accB = (s & 1) ? pk_add_bc_hi(accB, xh[(s >> 1) - M0]) : pk_add_bc_lo(accB, xh[(s >> 1) - M0]);
        if (((S + 2) >> 1) >= M0 && ((S + 2) >> 1) < M1) accB.y += xh[((S + 2) >> 1) - M0].y;
#define MAREX_XC(i, dst) \
    if (((H + i) >> 1) >= M0 && ((H + i) >> 1) < M1) dst = ((H + i) & 1) ? xh[((H + i) >> 1) - M0].y : xh[((H + i) >> 1) - M0].x;
        MAREX_XC(0, xc0) MAREX_XC(1, xc1) MAREX_XC(2, xc2) MAREX_XC(3, xc3)
#undef MAREX_XC
    };
    // mid() runs between the second half's loads and its sums
    auto smooth = [&](const int4& cA, v2f& smA, v2f& smB, v2f& xcA, v2f& xcB, int how, int ts) __attribute__((always_inline)) {
        load_half(std::integral_constant<int, 0>{}, how, ts);
        half(std::integral_constant<int, 0>{});
        load_half(std::integral_constant<int, 1>{}, how, ts);
        mid(cA);
        half(std::integral_constant<int, 1>{});
        smA = div_const2(accA, Sf, yS);
        smB = div_const2(accB, Sf, yS);
        xcA = (v2f){xc0, xc1};
        xcB = (v2f){xc2, xc3};
    };
    constexpr bool recip_exact = (W & 1) || (W & (W - 1)) == 0;
    auto digit2 = [&](v2f a, int& k0, int& k1) __attribute__((always_inline)) {  // np.digitize - 1 on the arange table (see k_shift_fast)
        const v2f f = __builtin_elementwise_fma(a, splat2(inv_width), splat2(c0));
        v2f t;
        t.x = __builtin_amdgcn_fmed3f(__builtin_floorf(f.x), 0.0f, nbm1f);
        t.y = __builtin_amdgcn_fmed3f(__builtin_floorf(f.y), 0.0f, nbm1f);
        const v2f phi = t * splat2(e_delta);
        const v2f ehi = splat2(e_first) + phi;
        k0 = (int)t.x + (a.x >= ehi.x ? 1 : 0);
        k1 = (int)t.y + (a.y >= ehi.y ? 1 : 0);
    };
    // climatology of the W entries from J on + the anomaly; nanmean where the plain mean is NaN while the centre value is a number
    auto anomaly = [&](auto Jc, v2f xcA, v2f xcB, v2f& aA, v2f& aB) __attribute__((always_inline)) {
        constexpr int J = decltype(Jc)::value;
        v2f sA = splat2(0.f), sB = splat2(0.f);
#pragma unroll
        for (int j = 0; j < W; ++j) {
            sA = sA + hA[J + j];
            sB = sB + hB[J + j];
        }
        v2f climA, climB;
        if (recip_exact) {
            climA = div_const2(sA, Wf, yW);
            climB = div_const2(sB, Wf, yW);
        } else {
            climA = (v2f){sA.x / Wf, sA.y / Wf};
            climB = (v2f){sB.x / Wf, sB.y / Wf};
        }
        const bool slow = (!(climA.x == climA.x) && xcA.x == xcA.x) || (!(climA.y == climA.y) && xcA.y == xcA.y) ||
                          (!(climB.x == climB.x) && xcB.x == xcB.x) || (!(climB.y == climB.y) && xcB.y == xcB.y);
        if (__builtin_expect(__builtin_amdgcn_ballot_w64(slow) != 0, 0)) {
            float acc[4] = {0.f, 0.f, 0.f, 0.f};
            int n[4] = {0, 0, 0, 0};
#pragma unroll
            for (int j = 0; j < W; ++j) {
                const float v[4] = {hA[J + j].x, hA[J + j].y, hB[J + j].x, hB[J + j].y};
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if (v[i] == v[i]) {
                        acc[i] += v[i];
                        ++n[i];
                    }
            }
            if (!(climA.x == climA.x)) climA.x = acc[0] / (float)n[0];
            if (!(climA.y == climA.y)) climA.y = acc[1] / (float)n[1];
            if (!(climB.x == climB.x)) climB.x = acc[2] / (float)n[2];
            if (!(climB.y == climB.y)) climB.y = acc[3] / (float)n[3];
        }
        aA = xcA - climA;
        aB = xcB - climB;
    };

    // ---- a REGULAR year (all 4 dayofyears present, staged, no edge, one bucket position): no look at the calendar
    auto lean_year = [&](auto Jc, const int4& cA, const int4& cB) __attribute__((always_inline)) {
        constexpr int J = decltype(Jc)::value;
        v2f smA, smB, xcA, xcB;
        smooth(cA, smA, smB, xcA, xcB, 0, 0);
        // A wave whose 64 cells are all NaN on the 4 centre rows (land: 16 % of the cell groups of the 0.25-degree benchmark
        // field) has nothing to compute this year: the anomalies are the NaNs themselves, no key is counted.  Decided on the
        // values, year by year (a cell that is NaN today may hold numbers next year).
        const bool dead = __builtin_amdgcn_ballot_w64(xcA.x == xcA.x || xcA.y == xcA.y || xcB.x == xcB.x || xcB.y == xcB.y) == 0;
        if (dead) {
            n_invalid += 4;
            if (cA.x & LR_OUT) {
                if (!BINS) {
                    newkeys[wave][0][t_slot][lane] = 0u;
                    newkeys[wave][1][t_slot][lane] = 0u;
                    ++t_slot;
                }
                const unsigned long long ooff = ((unsigned long long)(unsigned)cB.y << 32) | (unsigned)cB.x;
                const rsrc_t ro = make_rsrc(reinterpret_cast<char*>(out) + ooff);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if (active) {  // NaN - climatology = the (quieted) NaN itself
                    stb_f32(ro, voff, 0, __builtin_canonicalizef(xcA.x));
                    stb_f32(ro, voff, rowb, __builtin_canonicalizef(xcA.y));
                    stb_f32(ro, voff, 2 * rowb, __builtin_canonicalizef(xcB.x));
                    stb_f32(ro, voff, 3 * rowb, __builtin_canonicalizef(xcB.y));
                    if (BINS) {  // NaN: the overflow bin
                        stb_u16(rbins, bin_lane, cB.z * 32, nb);
                        stb_u16(rbins, bin_lane, (cB.z + bD1) * 32, nb);
                        stb_u16(rbins, bin_lane, (cB.z + bD2) * 32, nb);
                        stb_u16(rbins, bin_lane, (cB.z + bD3) * 32, nb);
                    }
                }
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
        } else {
        n_invalid += (finite_f(xcA.x) ? 0 : 1) + (finite_f(xcA.y) ? 0 : 1) + (finite_f(xcB.x) ? 0 : 1) + (finite_f(xcB.y) ? 0 : 1);
        if (cA.x & LR_OUT) {
            v2f aA, aB;
            anomaly(Jc, xcA, xcB, aA, aB);
            int k0, k1, k2, k3;
            digit2(aA, k0, k1);
            digit2(aB, k2, k3);
            if constexpr (BINS) {
                k0 = (aA.x == aA.x) ? k0 : nb;  // NaN: the overflow bin (a value beyond the table got it from the digitize)
                k1 = (aA.y == aA.y) ? k1 : nb;
                k2 = (aB.x == aB.x) ? k2 : nb;
                k3 = (aB.y == aB.y) ? k3 : nb;
                const unsigned long long ooffb = ((unsigned long long)(unsigned)cB.y << 32) | (unsigned)cB.x;
                const rsrc_t rob = make_rsrc(reinterpret_cast<char*>(out) + ooffb);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if (active) {
                    stb_f32(rob, voff, 0, aA.x);
                    stb_f32(rob, voff, rowb, aA.y);
                    stb_f32(rob, voff, 2 * rowb, aB.x);
                    stb_f32(rob, voff, 3 * rowb, aB.y);
                    stb_u16(rbins, bin_lane, cB.z * 32, k0);
                    stb_u16(rbins, bin_lane, (cB.z + bD1) * 32, k1);
                    stb_u16(rbins, bin_lane, (cB.z + bD2) * 32, k2);
                    stb_u16(rbins, bin_lane, (cB.z + bD3) * 32, k3);
                }
            } else {
            const unsigned pos = (unsigned)cB.z, kb = 128u + pos;  // ((k + 1) << 7) | pos = (k << 7) + (128 + pos)
            const bool v0 = aA.x < e_last, v1 = aA.y < e_last, v2 = aB.x < e_last, v3 = aB.y < e_last;  // countable: a number below the last edge
            const unsigned q0 = v0 ? ((unsigned)k0 << TAIL_POS_BITS) + kb : 0u, q1 = v1 ? ((unsigned)k1 << TAIL_POS_BITS) + kb : 0u;
            const unsigned q2 = v2 ? ((unsigned)k2 << TAIL_POS_BITS) + kb : 0u, q3 = v3 ? ((unsigned)k3 << TAIL_POS_BITS) + kb : 0u;
            newkeys[wave][0][t_slot][lane] = q0 | (q1 << 16);
            newkeys[wave][1][t_slot][lane] = q2 | (q3 << 16);
            t_cntA += (v0 ? 1u : 0u) + (v1 ? 0x10000u : 0u);
            t_cntB += (v2 ? 1u : 0u) + (v3 ? 0x10000u : 0u);
            const float mx = fmaxf(fmaxf(aA.x, aA.y), fmaxf(aB.x, aB.y));  // maxNum skips NaN
            if (__builtin_expect(__builtin_amdgcn_ballot_w64(mx >= e_last) != 0, 0)) {  // values beyond the table: positions into the aux word
                unsigned s0 = t_ovfA & 0xFFFFu, s1 = t_ovfA >> 16, s2 = t_ovfB & 0xFFFFu, s3 = t_ovfB >> 16;
                if (aA.x >= e_last) s0 = tail_ovf_add(s0, pos);
                if (aA.y >= e_last) s1 = tail_ovf_add(s1, pos);
                if (aB.x >= e_last) s2 = tail_ovf_add(s2, pos);
                if (aB.y >= e_last) s3 = tail_ovf_add(s3, pos);
                t_ovfA = s0 | (s1 << 16);
                t_ovfB = s2 | (s3 << 16);
            }
            ++t_slot;
            const unsigned long long ooff = ((unsigned long long)(unsigned)cB.y << 32) | (unsigned)cB.x;
            const rsrc_t ro = make_rsrc(reinterpret_cast<char*>(out) + ooff);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // next year's rows have landed; the stores below stay in flight
            if (active) {  // lanes beyond C hold other cells' rows (DMA): they store nothing
                stb_f32(ro, voff, 0, aA.x);
                stb_f32(ro, voff, rowb, aA.y);
                stb_f32(ro, voff, 2 * rowb, aB.x);
                stb_f32(ro, voff, 3 * rowb, aB.y);
            }
            }
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        }
        hA[J + W] = smA;  // year y joins the history
        hB[J + W] = smB;
    };

    // ---- any other year (k_shift_fast's body): first days of the series, the leap-day chunk, trailing partial years, waves
    // that only help staging
    auto general_year = [&](auto Jc, int y, const int4& cA) __attribute__((always_inline)) {
        constexpr int J = decltype(Jc)::value;
        v2f smA = splat2(qnan), smB = splat2(qnan);
        int4 pA = make_int4(-1, -1, 0, 0), pB = make_int4(0, 0, 0, 0);
        if (mine) {
            pA = fplan[((size_t)y * 92 + chunk) * 2];
            pB = fplan[((size_t)y * 92 + chunk) * 2 + 1];
        }
        const int nd = pA.z & 0xFF;
        if (mine && nd > 0) {
            const int tb = cA.y, r0 = pA.x - H;
            const bool edge = r0 < 0 || r0 + 2 * NPAIR > T;
            const bool staged = tb >= H && tb - H + NROWS <= T && pA.x == tb + 4 * wave;
            v2f xcA, xcB;
            smooth(cA, smA, smB, xcA, xcB, staged ? 0 : (edge ? 2 : 1), pA.x);
            if (edge) {  // windows that leave the series: NaN
                const int t0 = pA.x;
                smA.x = (t0 - H >= 0 && t0 + H < T) ? smA.x : qnan;
                smA.y = (t0 + 1 - H >= 0 && t0 + 1 + H < T) ? smA.y : qnan;
                smB.x = (t0 + 2 - H >= 0 && t0 + 2 + H < T) ? smB.x : qnan;
                smB.y = (t0 + 3 - H >= 0 && t0 + 3 + H < T) ? smB.y : qnan;
            }
            const bool has1 = nd > 1, has2 = nd > 2, has3 = nd > 3;
            n_invalid += finite_f(xcA.x) ? 0 : 1;
            if (has1) n_invalid += finite_f(xcA.y) ? 0 : 1; else smA.y = qnan;
            if (has2) n_invalid += finite_f(xcB.x) ? 0 : 1; else smB.x = qnan;
            if (has3) n_invalid += finite_f(xcB.y) ? 0 : 1; else smB.y = qnan;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (pA.y >= 0) {  // output rows
                v2f aA, aB;
                anomaly(Jc, xcA, xcB, aA, aB);
                const rsrc_t ro = make_rsrc(out + (size_t)pA.y * (size_t)C);
                if (active) {
                    stb_f32(ro, voff, 0, aA.x);
                    if (has1) stb_f32(ro, voff, rowb, aA.y);
                    if (has2) stb_f32(ro, voff, 2 * rowb, aB.x);
                    if (has3) stb_f32(ro, voff, 3 * rowb, aB.y);
                }
                int k0, k1, k2, k3;
                digit2(aA, k0, k1);
                digit2(aB, k2, k3);
                if constexpr (BINS) {
                    if (active) {
                        stb_u16(rbins, bin_lane, pA.w * 32, (aA.x == aA.x) ? k0 : nb);
                        if (has1) stb_u16(rbins, bin_lane, pB.x * 32, (aA.y == aA.y) ? k1 : nb);
                        if (has2) stb_u16(rbins, bin_lane, pB.y * 32, (aB.x == aB.x) ? k2 : nb);
                        if (has3) stb_u16(rbins, bin_lane, pB.z * 32, (aB.y == aB.y) ? k3 : nb);
                    }
                } else {
                const float l1 = has1 ? e_last : -__builtin_inff(), l2 = has2 ? e_last : -__builtin_inff(),
                            l3 = has3 ? e_last : -__builtin_inff();  // uniform
                const bool v0 = aA.x < e_last, v1 = aA.y < l1, v2 = aB.x < l2, v3 = aB.y < l3;
                const int ds0 = tails.doy_start[d0], ds1 = tails.doy_start[d0 + 1 < NDOY ? d0 + 1 : NDOY - 1];
                const int ds2 = tails.doy_start[d0 + 2 < NDOY ? d0 + 2 : NDOY - 1], ds3 = tails.doy_start[d0 + 3 < NDOY ? d0 + 3 : NDOY - 1];
                const unsigned p0 = (unsigned)(pA.w - ds0), p1 = (unsigned)(pB.x - ds1), p2 = (unsigned)(pB.y - ds2), p3 = (unsigned)(pB.z - ds3);
                const unsigned q0 = v0 ? ((unsigned)k0 << TAIL_POS_BITS) + 128u + p0 : 0u, q1 = v1 ? ((unsigned)k1 << TAIL_POS_BITS) + 128u + p1 : 0u;
                const unsigned q2 = v2 ? ((unsigned)k2 << TAIL_POS_BITS) + 128u + p2 : 0u, q3 = v3 ? ((unsigned)k3 << TAIL_POS_BITS) + 128u + p3 : 0u;
                newkeys[wave][0][t_slot][lane] = q0 | (q1 << 16);
                newkeys[wave][1][t_slot][lane] = q2 | (q3 << 16);
                t_cntA += (v0 ? 1u : 0u) + (v1 ? 0x10000u : 0u);
                t_cntB += (v2 ? 1u : 0u) + (v3 ? 0x10000u : 0u);
                const float mx = fmaxf(fmaxf(aA.x, has1 ? aA.y : aA.x), fmaxf(has2 ? aB.x : aA.x, has3 ? aB.y : aA.x));
                if (__builtin_amdgcn_ballot_w64(mx >= e_last) != 0) {
                    unsigned s0 = t_ovfA & 0xFFFFu, s1 = t_ovfA >> 16, s2 = t_ovfB & 0xFFFFu, s3 = t_ovfB >> 16;
                    if (aA.x >= e_last) s0 = tail_ovf_add(s0, p0);
                    if (has1 && aA.y >= e_last) s1 = tail_ovf_add(s1, p1);
                    if (has2 && aB.x >= e_last) s2 = tail_ovf_add(s2, p2);
                    if (has3 && aB.y >= e_last) s3 = tail_ovf_add(s3, p3);
                    t_ovfA = s0 | (s1 << 16);
                    t_ovfB = s2 | (s3 << 16);
                }
                ++t_slot;
                }
            }
        } else {
            mid(cA);  // nothing to compute: the barrier, the staging and the record prefetch all the same
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        hA[J + W] = smA;
        hB[J + W] = smB;
    };

    // (a lambda around the year makes hipcc keep a closure -- and with it a history line -- in scratch memory: a macro it is)
#define MAREX_LEAN_YEAR(J, yy)                                                                                  \
    {                                                                                                           \
        const int4 cA = nA, cB = nB;                                                                            \
        if (mine && (cA.x & LR_REG)) {                                                                          \
            lean_year(std::integral_constant<int, J>{}, cA, cB);                                                \
            ++n_lean;                                                                                           \
        } else {                                                                                                \
            general_year(std::integral_constant<int, J>{}, yy, cA);                                             \
            n_gen += mine ? 1 : 0;                                                                              \
        }                                                                                                       \
        asm volatile("s_barrier" ::: "memory"); /* every wave's DMA has landed: the stage holds next year's rows */ \
        if (!BINS && t_slot == SHIFT_LIST) tails_flush();                                                       \
    }
    for (int y = 0; y < n_cal; y += U) {
        MAREX_LEAN_YEAR(0, y)
        if constexpr (U > 1) if (y + 1 < n_cal) MAREX_LEAN_YEAR(1, y + 1)
        if constexpr (U > 2) if (y + 2 < n_cal) MAREX_LEAN_YEAR(2, y + 2)
        if constexpr (U > 3) if (y + 3 < n_cal) MAREX_LEAN_YEAR(3, y + 3)
#pragma unroll
        for (int j = 0; j < W; ++j) {
            hA[j] = hA[j + U];
            hB[j] = hB[j + U];
        }
    }
#undef MAREX_LEAN_YEAR
    if (mine && !BINS) {
        if (t_slot > 0) tails_flush();
        while (t_list < tails.nper) tails_flush();
        if (active) {
            const unsigned cn[4] = {t_cntA & 0xFFFFu, t_cntA >> 16, t_cntB & 0xFFFFu, t_cntB >> 16};
            const unsigned ov[4] = {t_ovfA & 0xFFFFu, t_ovfA >> 16, t_ovfB & 0xFFFFu, t_ovfB >> 16};
#pragma unroll
            for (int di = 0; di < 4; ++di)
                if (d0 + di < NDOY) tails.aux[(size_t)(d0 + di) * (size_t)C + c] = tail_aux_word(cn[di], ov[di]);
        }
    }
    if (invalid_count && active && n_invalid) atomicAdd(&invalid_count[c], n_invalid);
#ifdef MAREX_LEAN_COUNTERS  // diagnostic builds only: two atomics per wave on two addresses cost 4 ms on a 10-yr field (750 000 waves)
    if (tails.dbg && lane == 0) {
        atomicAdd(&tails.dbg[6], (unsigned long long)n_lean);
        atomicAdd(&tails.dbg[7], (unsigned long long)n_gen);
    }
#endif
}

static bool lean_instance(int W, int S) { return S == 21 && (W == 15 || W == 5); }

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
    bool lean = false;  // k_shift_lean takes the tails configuration
    const int4* lplan = nullptr;  // its lean records
    int lean_waves = 4;           // waves per workgroup of k_shift_lean
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
#define MAREX_SF_ARGS dim3(xcd_grid(ncg, 23)), dim3(256), 0, ctx->stream, a.x, (long)a.T, (long)a.C, a.fplan, a.n_cal, a.skip, a.write_clim, a.edges, a.nb, (long)a.T_out, a.out, a.bins, a.mask, a.invalid_count, ncg, 23, a.tails
    if (a.lean && (a.tails.lists || a.bins)) {
        if constexpr (S == 21 && (W == 15 || W == 5)) {
            // (NWV = 8 -- 52 rows staged for 32 dayofyears -- measured 11.2 ms against 10.0 per 100-yr band and is not instantiated)
            if (a.tails.lists)
                hipLaunchKernelGGL((k_shift_lean<W, S, 4, false>), dim3(xcd_grid(ncg, 23)), dim3(256), 0, ctx->stream, a.x, (int)a.T, (int)a.C,
                                   a.fplan, a.lplan, a.n_cal, a.skip, a.edges, a.nb, a.out, a.mask, a.invalid_count, ncg, 23, a.tails,
                                   (unsigned short*)nullptr, (long)a.T_out);
            else
                hipLaunchKernelGGL((k_shift_lean<W, S, 4, true>), dim3(xcd_grid(ncg, 23)), dim3(256), 0, ctx->stream, a.x, (int)a.T, (int)a.C,
                                   a.fplan, a.lplan, a.n_cal, a.skip, a.edges, a.nb, a.out, a.mask, a.invalid_count, ncg, 23, a.tails,
                                   a.bins, (long)a.T_out);
        }
    } else if (a.tails.lists) {
        // LDS-DMA staging + L2 prefetch (option SHIFT_DMA=1; measured round 3 on a 100-yr band: 12.17 ms against 11.84 ms with
        // the rows staged through VGPRs -- the kernel is bound by instruction issue, not by the latency of its loads -- so it
        // is off by default and instantiated for the two benchmark shapes only): whole 16-byte segments of a row are either
        // inside the field or beyond it (C a multiple of 4), and the prefetch offset of 365 rows fits the 32-bit buffer offset
        constexpr int RPW = (S + 15 + 3) / 4;
        constexpr bool has_dma = S == 21 && (W == 15 || W == 5);
        const bool dma = has_dma && ctx_opt(ctx, "SHIFT_DMA", 0) != 0 && a.C >= 4 && a.C % 4 == 0 &&
                         (unsigned long long)(365 + 4 * RPW + 4) * (unsigned long long)a.C * 4ull < 0xFFFFFFFFull;
        if (dma)
            hipLaunchKernelGGL((k_shift_fast<W, true, S, has_dma>), MAREX_SF_ARGS);
        else
            hipLaunchKernelGGL((k_shift_fast<W, true, S, false>), MAREX_SF_ARGS);
    } else {
        hipLaunchKernelGGL((k_shift_fast<W, false, S, false>), MAREX_SF_ARGS);
    }
#undef MAREX_SF_ARGS
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
        // the lean kernel stages its rows by LDS-DMA: whole 16-byte segments of a row inside the field or beyond it (C % 4 == 0)
        // (the bin-matrix variant, option SHIFT_LEAN_BINS: short series whose thresholds come from the band kernel)
        const bool lean_bins = !tails.lists && bins && !write_clim && ctx_opt(ctx, "SHIFT_LEAN_BINS", 1) != 0;
        const bool lean = (tails.lists || lean_bins) && lean_instance(W, S) && ctx_opt(ctx, "SHIFT_LEAN", 1) != 0 && C >= 4 && C % 4 == 0 &&
                          T < (1ll << 31) - 64 && (unsigned long long)(S + 31 + 8 + 4) * (unsigned long long)C * 4ull < 0xFFFFFFFFull;
        const int lean_waves = 4;
        if (lean && ctx->shift_lplan_years < (size_t)n_cal_years) {
            if (ctx->shift_lplan) {
                HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
                (void)hipFree(ctx->shift_lplan);
                ctx->shift_lplan = nullptr;
                ctx->shift_lplan_years = 0;
            }
            HIP_TRY(ctx, hipMalloc((void**)&ctx->shift_lplan, ((size_t)n_cal_years + 2) * LEAN_CHUNKS * 8 * sizeof(int)));
            ctx->shift_lplan_years = (size_t)n_cal_years;
        }
        hipLaunchKernelGGL(k_shift_classify, dim3(1), dim3(1024), 0, ctx->stream, a.year_plan, n_cal_years, edges, nb,
                           (bins || tails.lists) ? 1 : 0, 1, ctx->shift_info, reinterpret_cast<int4*>(ctx->shift_plan), (long)T, (long)C,
                           lean ? S : 0, lean ? reinterpret_cast<int4*>(ctx->shift_lplan) : nullptr, tails.doy_start, lean_waves,
                           (lean && lean_bins) ? 1 : 0);
        a.lean_waves = lean_waves;
        a.lplan = reinterpret_cast<const int4*>(ctx->shift_lplan);
        a.lean = lean;
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
                            uint32_t* aux, const int* skip_chunks);

extern "C" int marex_shifting_baseline_tails_f32(marex_ctx* ctx, const float* x, int64_t T, int64_t C, const int32_t* year_plan,
                                                 int n_cal_years, int W, int S, const float* edges, int nb, int64_t T_out,
                                                 float* out, uint8_t* mask, int32_t* invalid_count, const int32_t* doy_start,
                                                 const int32_t* doy_rows, int max_bucket, void* lists, uint32_t* aux) {
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
