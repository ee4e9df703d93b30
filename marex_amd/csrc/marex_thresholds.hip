// marex_thresholds.hip -- K_T: pooled day-of-year histogram quantile
#include "marex_common.hip.h"

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
#ifndef TB_ROWWIN
#define TB_ROWWIN 16  // levels a tile row reads at once in the row-pooled search
#endif
#ifndef TB_ROWBACK
#define TB_ROWBACK 2  // how far below yesterday's level a lane wants its row window to start
#endif
#ifndef TB_BATCH
#define TB_BATCH 16
#endif
#define TB_BWF 64     // bins of the band that follows the thresholds
#define TB_MARGIN 6   // re-centre when the day's quantile bins come this close to a band edge
#define TB_STEP 56    // band shift when answering stragglers (8 bins of overlap)
static_assert(TB_STEP <= TB_BWF - 8, "straggler passes must overlap: a quantile bin between two tried bands would never be found");

template <int P, int TC, int NT, int TR_ = NT / TC>
__global__ void __launch_bounds__(NT, NT == 256 ? 4 : 1)
k_thr_band(const unsigned short* __restrict__ bins, long T_out, long C, int ny, int nx, int row0, int row1, int tiles_x,
           int Dd, int shift, int env_exact, int ablate, const int* __restrict__ doy_start,
           const float* __restrict__ first_anom, const float* __restrict__ centres, int nb, double q, int wd,
           float lower_bound, float upper_bound, float* __restrict__ thr, marex_thr_stats* __restrict__ stats,
           unsigned char* __restrict__ gscratch, int coarse_pd, int two_ended, int bwf) {
    // TR x TC tile cells on NT threads; a tile whose cell count is not a multiple of 64 (34 x 30 = 1020) leaves the last
    // lanes of the last wave without a cell: they own a private, unused column and take part in the barriers only
    constexpr int TR = TR_;
    constexpr int NCELL = TR * TC;
    static_assert(NCELL <= NT && NT - NCELL < 64, "tile does not match the thread count");
    constexpr int OR = TR - 2 * P, OC = TC - 2 * P;
#ifndef MAREX_ABLATION
    ablate = 0;  // timing-only ablation bits: -DMAREX_ABLATION builds only
#endif
    // lane-major level columns: TB_LS dwords (= 68 uint16 levels) per lane.  The stride 34 keeps 8-byte
    // alignment and makes 8-byte accesses of 32 consecutive lanes hit 64 distinct banks.
    __shared__ unsigned lev[NT * TB_LS];
    // per (day of the block, lane) state byte in a global scratch slab (every thread touches only its own bytes, a few
    // per day): keeping it out of LDS lets four 256-thread tiles share a CU and lifts the limit on Dd for the big tiles
    unsigned char (*gst)[NT] = reinterpret_cast<unsigned char (*)[NT]>(
        gscratch + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * (size_t)Dd * NT);
    __shared__ int s_gmin, s_gmax, s_unres;
    __shared__ unsigned tot_s[NT];  // per tile cell: number of samples in its window (top of the cumulative column)
    __shared__ int s_lo[2], s_hi[2];        // per pass (parity): some output's quantile lies below / above the band
    __shared__ int s_iumin[2], s_iumax[2];  // per day (parity): range of the quantile bins found

    const int t = threadIdx.x;
    // lane -> tile cell: rows are rotated by P so that the output rows P .. TR-P-1 fill the FIRST waves completely and
    // the halo rows share the last one(s), which then skip the per-output phase (a 16x16 tile with P = 2: three waves
    // at 48/64 output lanes + one idle, instead of 24+48+48+24 over four)
    const bool spare = NCELL < NT && t >= NCELL;
    const int tc = t % TC;
    const int tr = (TR > 1) ? (t / TC + P) % TR : 0;
    const int ci = spare ? t : tr * TC + tc;  // column / total slot of this lane
    const int ty = (int)blockIdx.x / tiles_x, tx = (int)blockIdx.x - ty * tiles_x;
    const int jt0 = row0 + ty * OR, it0 = tx * OC;
    const int j = (ny > 0) ? jt0 - P + tr : 0;
    const int icol = it0 - P + tc;
    bool cell_valid;
    long cell;
    if (ny > 0) {
        int gi = icol % nx;
        if (gi < 0) gi += nx;
        cell_valid = (j >= 0 && j < ny) && !spare;
        cell = (long)j * nx + gi;
    } else {
        cell_valid = icol < nx;
        cell = icol;
    }
    const bool is_out = !spare && tr >= P && tr < TR - P && tc >= P && tc < TC - P && j < row1 && icol < nx;
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

    unsigned* mycol = &lev[ci * TB_LS];  // columns are indexed by tile cell: neighbour offsets stay linear
    uint2* mycol2 = reinterpret_cast<uint2*>(mycol);
    for (int r = 0; r < TB_LS / 2; ++r) mycol2[r] = make_uint2(0u, 0u);
    if (t == 0) {
        s_gmin = 255;
        s_gmax = -1;
        s_unres = env_exact;
    }

    // level mapping of the current pass: level(b) = clamp((b >> lsh) + loff, 0, lhi)
    //   coarse: lsh = shift, loff = 0, lhi = ngroups - 1;  fine band B0..B0+BW-1: lsh = 0, loff = 1 - B0, lhi = BW + 1
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
            for (int u = 0; u < TB_PRE; ++u)
                if (u < pr.nd) bump(pr.b[u], sgn);  // uniform: a 5-sample bucket (10-yr series) is 5 updates, not 8
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
    // ---- TC == 16: a tile row is a DPP row of the wave.  The pooled count of a lane is then a VERTICAL sum in its own tile
    // column (2P+1 reads) followed by a HORIZONTAL sum across the row with row_shr / row_shl (zero beyond the row: halo
    // columns never ask) -- 5 + 4 instead of 25 reads and adds per dword, provided the lanes of a row look at the same levels.
    constexpr bool ROWPOOL = TC == 16 && TR > 1 && P >= 1 && P <= 3 && NT == 256;
    auto row_sum = [&](unsigned v) -> unsigned {
        unsigned h = v;
        if constexpr (P >= 1) {
            h += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xF, 0xF, true);
            h += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x101, 0xF, 0xF, true);
        }
        if constexpr (P >= 2) {
            h += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xF, 0xF, true);
            h += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x102, 0xF, 0xF, true);
        }
        if constexpr (P >= 3) {
            h += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x113, 0xF, 0xF, true);
            h += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x103, 0xF, 0xF, true);
        }
        return h;
    };
    auto row_min = [&](int m) -> int {  // minimum over the 16 lanes of the row, in every lane
        int o;
        o = __builtin_amdgcn_update_dpp(m, m, 0x128, 0xF, 0xF, false); m = o < m ? o : m;
        o = __builtin_amdgcn_update_dpp(m, m, 0x124, 0xF, 0xF, false); m = o < m ? o : m;
        o = __builtin_amdgcn_update_dpp(m, m, 0x122, 0xF, 0xF, false); m = o < m ? o : m;
        o = __builtin_amdgcn_update_dpp(m, m, 0x121, 0xF, 0xF, false); m = o < m ? o : m;
        return m;
    };
    // tile cell whose column a lane sums vertically: its own, rows clamped into the part of the tile that has 2P+1 rows around
    // it (lanes of halo rows compute something nobody reads)
    const int trs = tr < P ? P : (tr > TR - 1 - P ? TR - 1 - P : tr);
    const int cis = trs * TC + tc;
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
        if (ablate & 16) {  // timing only: no table look-ups, no float64 interpolation
            thr[(size_t)d * C + cell] = (float)(iu + ck + cb);
            return;
        }
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
    //   mode 2  fine levels, exact band g_base..: resolve the output-days whose group lies in the band
    //   (mode 1, the speculative fixed band of rounds 1-3, is gone: `follow` below walks the block with a band that moves)
    // init_pd < pd: day 0 of a coarse sweep sees only the 2*init_pd+1 central buckets -- good enough to PLACE the
    // speculative band (a wrong guess only sends the block to the exact path), never used for a result
    auto sweep = [&](int mode, int nd_pass, int ng, int init_pd, int d_off = 0) {  // d_off: first day of the sweep within the block
        if (mode == 0) {
            nlev = ngroups;
            lsh = shift;
            loff = 0;
            lhi = ngroups - 1;
        } else {
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
            const int dg = dd + d_off;  // day of the block (state bytes are per block day)
            const int d = d_begin + dg;
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
            if (!(ablate & 2)) tot_s[ci] = prefix(nlp);
            if (dd + 1 < nd_pass && !(ablate & 4)) {
                pin = load_bucket((d + 1 + pd) % NDOY);
                pout = load_bucket(((d - pd) % NDOY + NDOY) % NDOY);
            }
            __syncthreads();
            // ---------------- P2: quantile level of this lane's output cell
            const int g = (ablate & 1) ? 255 : gst[dg][t];
            if (mode == 0) {
                int g_lo = 255, g_hi = -1;
                if (g == 254) {
                    const int tot = pooled_tot();
                    if (tot > 0) {
                        int ck, cb;
                        int gg = find_level(hint, 0, nlev, q * (double)tot, false, ck, cb);
                        if (gg >= nlev) gg = nlev - 1;  // nothing above qpos: iu clips to nb-1
                        hint = gg;
                        gst[dg][t] = (unsigned char)gg;
                        g_lo = g_hi = gg;
                    } else if (init_pd == pd) {
                        gst[dg][t] = 255;
                        thr[(size_t)d * C + cell] = nan_f();  // empty window
                    }
                }
                if (__builtin_amdgcn_ballot_w64(g_hi >= 0) != 0) {  // one pair of LDS atomics per wave (64 lanes on one address are 64 serial operations)
                    const int wlo = wave_min_i32(g_lo), whi = wave_max_i32(g_hi);
                    if ((t & 63) == 0) {
                        atomicMin(&s_gmin, wlo);
                        atomicMax(&s_gmax, whi);
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
                    gst[dg][t] = 255;
                } else {
                    hint = -1;
                }
            }
            __syncthreads();
        }
    };

    // ---- the band FOLLOWS the thresholds (round 4; the logic of k_thr_tails on the bin matrix): one sweep over the block's days
    // with one level per bin of a 64-bin band.  After every day the band is re-centred on the day's range of quantile bins
    // whenever that range comes within TB_MARGIN bins of an edge (the columns are then rebuilt from the wd buckets of the next
    // day: cheap, a bucket is a handful of samples per cell on this path), and an output whose quantile lies outside the band
    // reports the direction: the tile rebuilds its columns around a band further down / up and answers the stragglers in extra
    // passes of the same day.  Seasonal drift costs a rebuild every few weeks of the walk -- the speculative band of round 1
    // sent the whole block through two or three more sweeps (the exact path) as soon as one output-day escaped.
    auto follow = [&](int nd_pass, int b0_first) {
        const int step = bwf - 8;  // straggler bands overlap by 8 bins
        const int base_max = nb - bwf > 0 ? nb - bwf : 0;
        auto clamp_b0 = [&](int b) { return b < 0 ? 0 : (b > base_max ? base_max : b); };
        auto set_band = [&](int b0) {
            B0 = clamp_b0(b0);
            BW = nb - B0 < bwf ? nb - B0 : bwf;
            nlev = BW + 2;
            lsh = 0;
            loff = 1 - B0;
            lhi = BW + 1;
        };
        auto wrapd = [&](int d) { return ((d % NDOY) + NDOY) % NDOY; };
        auto rebuild = [&](int d) {  // this lane's column of day d from scratch, current band (not yet prefix-summed)
            for (int r = 0; r < TB_LS / 2; ++r) mycol2[r] = make_uint2(0u, 0u);
            Pre cur = load_bucket(wrapd(d - pd));
            for (int o = -pd + 1; o <= pd; ++o) {
                const Pre nxt = load_bucket(wrapd(d + o));
                apply_bucket(cur, +1);
                cur = nxt;
            }
            apply_bucket(cur, +1);
        };
        set_band(b0_first);
        if (t == 0) {
            s_lo[0] = s_lo[1] = s_hi[0] = s_hi[1] = 0;
            s_iumin[0] = s_iumin[1] = 0x7fffffff;
            s_iumax[0] = s_iumax[1] = -1;
        }
        __syncthreads();
        const int nlp = (bwf + 2 + 1) >> 1;
        const int pass_limit = 2 * (nb / step) + 9;
        int hint = -1;
        bool need_rebuild = true;
        Pre pin, pout;
        for (int dd = 0; dd < nd_pass; ++dd) {
            const int d = d_begin + dd, dpar = dd & 1;
            // ---------------- P1
            if (need_rebuild) {
                rebuild(d);
                need_rebuild = false;
                hint = -1;
            } else {
                unprefix(nlp);
                apply_bucket(pin, +1);
                apply_bucket(pout, -1);
            }
            tot_s[ci] = prefix(nlp);
            if (dd + 1 < nd_pass) {
                pin = load_bucket(wrapd(d + 1 + pd));
                pout = load_bucket(wrapd(d - pd));
            }
            // ---------------- P2 (+ straggler passes)
            bool resolved = !(is_out && !land);
            int tried_lo = B0, tried_hi = B0;
            bool excursion = false;
            for (int pass = 0;; ++pass) {
                const int par = pass & 1;
                __syncthreads();
                if (t == 0) {
                    s_lo[par ^ 1] = 0;
                    s_hi[par ^ 1] = 0;
                    if (pass == 0) {
                        s_iumin[dpar ^ 1] = 0x7fffffff;
                        s_iumax[dpar ^ 1] = -1;
                    }
                }
                const bool need = !resolved;
                int k = -1, ck = 0, cb = 0, tot = 0;
                double qpos = 0.0;
                if constexpr (ROWPOOL) {
                    if (__builtin_amdgcn_ballot_w64(need) != 0) {  // wave-uniform: all 64 lanes run the shared part
                        unsigned tv = 0;
#pragma unroll
                        for (int dr = -P; dr <= P; ++dr) tv += tot_s[cis + dr * TC];
                        tot = (int)row_sum(tv);
                        qpos = q * (double)tot;
                        const bool want = need && tot > 0;
                        const int khi = BW + 1;
                        int top = (khi - 1) & ~3;
                        top = top > 60 ? 60 : (top < 0 ? 0 : top);
                        int start = ((hint >= 0 ? hint : ((1 + khi) >> 1)) - TB_ROWBACK) & ~3;
                        start = start < 0 ? 0 : (start > top ? top : start);
                        constexpr int RW = TB_ROWWIN;
                        int u = row_min(want ? start : 0x7fffffff);
                        u = u > 68 - RW ? 68 - RW : u;
                        unsigned a[RW / 2];
#pragma unroll
                        for (int i = 0; i < RW / 2; ++i) a[i] = 0u;
                        {
                            const unsigned* base = &lev[cis * TB_LS] + (u >> 1);
#pragma unroll
                            for (int dr = -P; dr <= P; ++dr) {
                                const uint2* p2 = reinterpret_cast<const uint2*>(base + dr * TC * TB_LS);
#pragma unroll
                                for (int i = 0; i < RW / 4; ++i) {
                                    const uint2 w = p2[i];
                                    a[2 * i] += w.x;
                                    a[2 * i + 1] += w.y;
                                }
                            }
                        }
#pragma unroll
                        for (int i = 0; i < RW / 2; ++i) a[i] = row_sum(a[i]);
                        if (want) {
                            bool solved = false;
                            const int qf = (int)floor(qpos);
                            const int mm = (khi - u) < RW ? (khi - u) : RW;
                            int Wv[RW];
#pragma unroll
                            for (int i = 0; i < RW; ++i) Wv[i] = (int)((a[i >> 1] >> (16 * (i & 1))) & 0xFFFFu);
                            int n = 0;
#pragma unroll
                            for (int i = 0; i < RW; ++i) n += (i < mm) && (Wv[i] <= qf);
                            if (n == 0) {
                                if (u == 0) {
                                    k = 0;
                                    ck = Wv[0];
                                    solved = true;
                                }
                            } else if (n == mm) {
                                if (mm < RW) {
                                    k = khi;
#pragma unroll
                                    for (int i = 0; i < RW; ++i)
                                        if (i == mm - 1) cb = Wv[i];
                                    solved = true;
                                }
                            } else {
                                k = u + n;
#pragma unroll
                                for (int i = 0; i < RW; ++i) {
                                    if (i == n - 1) cb = Wv[i];
                                    if (i == n) ck = Wv[i];
                                }
                                solved = true;
                            }
                            if (!solved) k = find_level(hint, 1, khi, qpos, true, ck, cb);  // rare: the lane's own windows
                        }
                    }
                } else {
                    if (need) {
                        tot = pooled_tot();
                        if (tot > 0) {
                            qpos = q * (double)tot;
                            k = find_level(hint, 1, BW + 1, qpos, true, ck, cb);
                        }
                    }
                }
                int iu_lo = 0x7fffffff, iu_hi = -1;  // this lane's contribution to the day's range of quantile bins
                if (need) {
                    if (tot > 0) {
                        int iu = -1;
                        if (k >= 1 && k <= BW) {
                            iu = B0 + k - 1;
                            hint = k;
                        } else if (k > BW && B0 + BW >= nb) {  // nothing exceeds qpos (q == 1): searchsorted gives nb, clipped to nb - 1
                            iu = nb - 1;
                            k = iu - B0 + 1;
                            ck = pooled(k);
                            cb = pooled(k - 1);
                            hint = k;
                        } else if (k > BW) {
                            atomicOr(&s_hi[par], 1);
                            hint = -1;
                        } else {
                            atomicOr(&s_lo[par], 1);
                            hint = -1;
                        }
                        if (iu >= 0) {
                            emit_threshold(d, iu, ck, cb, qpos);
                            iu_lo = iu_hi = iu;
                            resolved = true;
                        }
                    } else {
                        thr[(size_t)d * C + cell] = nan_f();  // empty window
                        resolved = true;
                    }
                }
                if (__builtin_amdgcn_ballot_w64(iu_hi >= 0) != 0) {  // one pair of LDS atomics per wave, not per lane
                    const int wlo = wave_min_i32(iu_lo), whi = wave_max_i32(iu_hi);
                    if ((t & 63) == 0) {
                        atomicMin(&s_iumin[dpar], wlo);
                        atomicMax(&s_iumax[dpar], whi);
                    }
                }
                __syncthreads();
                const int lo = s_lo[par], hi = s_hi[par];
                if (!lo && !hi) break;
                if (pass >= pass_limit) {  // cannot happen while the tried bands overlap (static_assert above); counted, never silent
                    if (!resolved) {
                        thr[(size_t)d * C + cell] = nan_f();
                        atomicAdd(&stats->n_unresolved, 1u);
                    }
                    break;
                }
                // stragglers: a band further down (first) or further up than anything tried for this day
                if (lo) {
                    set_band(tried_lo - step);
                    tried_lo = B0;
                } else {
                    set_band(tried_hi + step);
                    tried_hi = B0;
                }
                excursion = true;
                rebuild(d);
                tot_s[ci] = prefix(nlp);  // same totals (band independent); columns are read after the barrier above
                hint = -1;
            }
            // ---------------- band of the next day
            if (dd + 1 < nd_pass) {
                const int imin = s_iumin[dpar], imax = s_iumax[dpar];
                if (imax >= 0) {
                    const int span = imax - imin + 1;
                    const bool near_edge = imin - B0 < TB_MARGIN || (B0 + BW - 1) - imax < TB_MARGIN;
                    if (excursion || near_edge) {
                        const int want = clamp_b0(span <= bwf - 2 * TB_MARGIN ? imin - (bwf - span) / 2 : imin - TB_MARGIN);
                        if (want != B0) {
                            set_band(want);
                            need_rebuild = true;
                        }
                    }
                }
            }
        }
    };

    __syncthreads();
    // placement: coarse quantile groups on the block's FIRST and LAST day -- thresholds drift with the season, and a band
    // placed on day 0 alone loses the days at the far end of a long block to the exact path (measured on a field whose p95
    // swings by 0.5 K over the year: 11.5 instead of 8.4 ms per 100-yr band with 61-day blocks)
    const int cpd = (coarse_pd >= 0 && coarse_pd < pd) ? coarse_pd : pd;
    sweep(0, 1, 0, cpd);
    const int gmin0 = s_gmin, gmax0 = s_gmax;
    __syncthreads();
    int gmin = gmin0, gmax = gmax0;
    if (!env_exact && gmax0 >= 0) {
        // first band: centred on the bins of the coarse groups day 0 needs (a bad start only costs passes, never a result)
        const int blo = gmin0 << shift, bhi = ((gmax0 + 1) << shift) - 1;
        const int span = bhi - blo + 1;
        follow(ndays, span <= bwf - 2 * TB_MARGIN ? blo - (bwf - span) / 2 : blo - TB_MARGIN);
    } else if (t == 0) {
        s_unres = 1;  // THR_EXACT_PATH (tests), or a block whose first day has no window at all
    }
    (void)two_ended;
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
    const int algo = ctx_opt(ctx, "THR_ALGO", 0);   // 0 auto, 1 force sliding histograms
    const int p = ws / 2;
    const bool band_ok = (1 << shift) <= 64 && p <= 3 && max_bucket > 0 && (int64_t)max_bucket * wd * ws * ws <= 65535;
    if (algo != 1 && band_ok) {
        // tile: 16x16 cells / 256 threads, or 32x32 / 1024 threads (less halo redundancy, more output lanes)
        // long dayofyear buckets (many years) make the kernel sample-streaming bound: the big tile re-streams
        // 1.31x instead of 1.78x halo cells per output cell (measured 17.4 vs 23.3 ms on an 85-year band)
        const int tile_pref = ctx_opt(ctx, "THR_TILE", max_bucket >= 24 ? 32 : 16);
        // tile_pref 32: 32x32 cells / 1024 threads (one workgroup per CU); 3216: 32 wide x 16 tall / 512 threads (two
        // independent workgroups per CU); 16: 16x16 / 256 threads
        const bool big = (ny > 0 && p > 0) && (tile_pref == 32 || tile_pref == 3216) && (row1 - row0) >= 16 && nx >= 16;
        const bool half = big && tile_pref == 3216;
        const int NT = big ? (half ? 512 : 1024) : 256;
        int TR = (ny > 0 && p > 0) ? (big ? (half ? 16 : 32) : 16) : 1, TC = NT / TR;
        // a 34 x 30 tile (1020 cells) covers the same area per workgroup as 32 x 32: take whichever needs fewer tiles
        // for the rows asked for (latitude bands of 90 rows: 3 x 56 tiles instead of 4 x 52, a fifth less work)
        bool tall = false;
        if (big && !half && p == 2 && ctx_opt(ctx, "THR_TALL", 1)) {
            auto ntiles = [&](int tr, int tc) {
                return (long)((nx + tc - 2 * p - 1) / (tc - 2 * p)) * ((row1 - row0 + tr - 2 * p - 1) / (tr - 2 * p));
            };
            tall = ntiles(34, 30) < ntiles(32, 32) && (row1 - row0) >= 30;
            if (tall) TR = 34, TC = 30;
        }
        const int OR = TR - 2 * p, OC = TC - 2 * p;
        const int tiles_x = (nx + OC - 1) / OC, tiles_y = (row1 - row0 + OR - 1) / OR;
        int Dd = ctx_opt(ctx, "THR_DD", 0);
        if (Dd < 1 || Dd > 128) {
            Dd = big ? 48 : TB_DMAX;
            if (big && !half) {
                // One 1024-thread tile per CU: the launch takes ceil(tiles * day-blocks / CUs) rounds of (Dd + wd - 1 + a few)
                // bucket-days each (window build-up + placement pass).  Pick the day-block length with the least total
                // -- e.g. 168 tiles on 256 CUs: 61 days (6 blocks, 3.94 rounds) beat 48 (8 blocks, 5.25 -> 6 rounds).
                const long cus = device_cus(ctx), tiles = (long)tiles_x * tiles_y;
                long best = -1;
                // (not beyond 64 days: the speculative band is placed on the block's first day, and real thresholds drift
                // with the season -- longer blocks would send more of them to the exact path)
                for (int d = 40; d <= 64; ++d) {
                    const long blocks = tiles * ((NDOY + d - 1) / d);
                    const long cost = ((blocks + cus - 1) / cus) * (d + wd + 3);
                    if (best < 0 || cost < best) best = cost, Dd = d;
                }
            }
        }
        dim3 grid((unsigned)(tiles_x * tiles_y), (unsigned)((NDOY + Dd - 1) / Dd));
        unsigned char* gscratch = nullptr;
        {  // per-(tile, day, lane) state bytes
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
        const int coarse_pd = ctx_opt(ctx, "THR_COARSE_PD", 1);
        int bwf = ctx_opt(ctx, "THR_BWF", TB_BWF);  // bins of the band that follows the thresholds (experiments: 24..64)
        bwf = bwf < 24 ? 24 : (bwf > TB_BWF ? TB_BWF : bwf);
#define MAREX_BAND_ARGS bins, (long)T_out, (long)C, ny, nx, row0, row1, tiles_x, Dd, shift, ctx_opt(ctx, "THR_EXACT_PATH", 0), MAREX_ABLATE_OPT(ctx, "THR_ABLATE"), doy_start, first_anom, centres, nb, q, wd, lower_bound, upper_bound, thr_doy_major, stats, gscratch, coarse_pd, ctx_opt(ctx, "THR_TWO_ENDS", 1), bwf
        {
            LaunchTimer lt(ctx, MAREX_K_THRESHOLDS);
            if (TR == 1)
                hipLaunchKernelGGL((k_thr_band<0, 256, 256>), grid, dim3(256), 0, ctx->stream, MAREX_BAND_ARGS);
            else if (half && p == 2)
                hipLaunchKernelGGL((k_thr_band<2, 32, 512>), grid, dim3(512), 0, ctx->stream, MAREX_BAND_ARGS);
            else if (big && p == 1)
                hipLaunchKernelGGL((k_thr_band<1, 32, 1024>), grid, dim3(1024), 0, ctx->stream, MAREX_BAND_ARGS);
            else if (big && p == 2 && tall)
                hipLaunchKernelGGL((k_thr_band<2, 30, 1024, 34>), grid, dim3(1024), 0, ctx->stream, MAREX_BAND_ARGS);
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
    const bool pack = max_bucket > 0 && (int64_t)max_bucket * wd * ws * ws <= 65535 && !ctx_opt(ctx, "THR_U32", 0);
    const int nbw = pack ? (nb + 1) / 2 : nb;
    int NW = ctx_opt(ctx, "THR_NW", 16);
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
