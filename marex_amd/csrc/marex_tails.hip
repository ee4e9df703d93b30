// marex_tails.hip -- K_X: tail extraction, K_T: day-of-year thresholds from tails, K_M: extreme mask from tails
#include "marex_common.hip.h"
#include "marex_tails.hip.h"

// ------------------------------------------------------------------------------------------------
// K_X: tails of an anomaly field (the counting stage of detect.py:2622-2648 in the form the threshold and mask
// kernels consume, see marex_tails.hip.h).  Thread = cell, two neighbouring dayofyears at a time as packed pairs:
// 16 rows of both buckets in flight, np.digitize, keys, a 16-key sorting network and a merge into the K best so far.
// Reads every anomaly once (coalesced 256-byte row segments per wave, like the mask kernel); the VALU work
// (~25 instructions per sample) hides under the HBM stream.
// ------------------------------------------------------------------------------------------------
template <int K>
__global__ void __launch_bounds__(256)
k_tail_extract(const float* __restrict__ anom, long C, const int* __restrict__ doy_start, const int* __restrict__ doy_rows,
               const float* __restrict__ edges, int nb, uint4* __restrict__ tails, unsigned short* __restrict__ aux) {
    extern __shared__ float e[];  // [nb + 1]
    constexpr int KC = K / 8;
    const int tid = threadIdx.x;
    for (int i = tid; i <= nb; i += 256) e[i] = edges[i];
    __syncthreads();
    const bool arange_tab = edges_are_arange(e, nb);
    const float inv_width = (float)(nb - 1) / (e[nb] - e[1]);
    const float e_first = e[1], e_delta = e[2] - e[1], e_last = e[nb];
    const long c = (long)blockIdx.x * 256 + tid;
    const bool active = c < C;
    const long cidx = active ? c : C - 1;
    const int npairs = NDOY / 2;
    const int pA = (int)blockIdx.y * npairs / (int)gridDim.y, pB = ((int)blockIdx.y + 1) * npairs / (int)gridDim.y;
    for (int p = pA; p < pB; ++p) {
        const int d0 = 2 * p, d1 = d0 + 1;
        const int s0 = doy_start[d0], n0 = doy_start[d0 + 1] - s0;
        const int s1 = doy_start[d1], n1 = doy_start[d1 + 1] - s1;
        const int nmax = n0 > n1 ? n0 : n1;
        unsigned top[K];
#pragma unroll
        for (int i = 0; i < K; ++i) top[i] = 0u;
        unsigned cnt0 = 0, cnt1 = 0, ovf0 = 0, ovf1 = 0;
        for (int r = 0; r < nmax; r += 16) {
            float va[16], vb[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const int pos = r + u;  // uniform
                va[u] = pos < n0 ? anom[(size_t)doy_rows[s0 + pos] * C + cidx] : nan_f();
                vb[u] = pos < n1 ? anom[(size_t)doy_rows[s1 + pos] * C + cidx] : nan_f();
            }
            unsigned nw[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const int pos = r + u;
                const int ba = arange_tab ? digitize_arange(va[u], e_first, e_delta, e_last, nb, inv_width) : digitize_bin(va[u], e, nb, inv_width);
                const int bb = arange_tab ? digitize_arange(vb[u], e_first, e_delta, e_last, nb, inv_width) : digitize_bin(vb[u], e, nb, inv_width);
                const unsigned ka = ba < nb ? tail_key(ba, pos) : 0u, kb = bb < nb ? tail_key(bb, pos) : 0u;
                cnt0 += ba < nb;
                cnt1 += bb < nb;
                ovf0 |= va[u] >= e_last;  // false for NaN
                ovf1 |= vb[u] >= e_last;
                nw[u] = ka | (kb << 16);
            }
            bitonic_sort_desc<16>(nw);
            tail_merge<K, 16>(top, nw);
        }
        if (active) {
#pragma unroll
            for (int j = 0; j < KC; ++j) {
                uint4 w0, w1;
                unsigned* a = reinterpret_cast<unsigned*>(&w0);
                unsigned* b = reinterpret_cast<unsigned*>(&w1);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const unsigned x = top[8 * j + 2 * i], y = top[8 * j + 2 * i + 1];
                    a[i] = (x & 0xFFFFu) | (y << 16);
                    b[i] = (x >> 16) | (y & 0xFFFF0000u);
                }
                tails[((size_t)d0 * KC + j) * C + c] = w0;
                tails[((size_t)d1 * KC + j) * C + c] = w1;
            }
            aux[(size_t)d0 * C + c] = (unsigned short)(cnt0 | (ovf0 ? 0x8000u : 0u));
            aux[(size_t)d1 * C + c] = (unsigned short)(cnt1 | (ovf1 ? 0x8000u : 0u));
        }
    }
}

extern "C" int marex_tail_extract_f32(marex_ctx* ctx, const float* anom, int64_t T_out, int64_t C, const int32_t* doy_start,
                                      const int32_t* doy_rows, int max_bucket, const float* edges, int nb, int K,
                                      void* tails, uint16_t* aux) {
    if (!ctx) return -1;
    if (!anom || !doy_start || !doy_rows || !edges || !tails || !aux || T_out <= 0 || C <= 0)
        return fail(ctx, -1, "marex_tail_extract_f32: null pointer or empty shape");
    if (nb < 4 || nb > TAIL_MAX_NB) return fail(ctx, -4, "marex_tail_extract_f32: nb must be in 4..%d", TAIL_MAX_NB);
    if (max_bucket < 1 || max_bucket > TAIL_MAX_BUCKET)
        return fail(ctx, -4, "marex_tail_extract_f32: dayofyear buckets must hold 1..%d rows", TAIL_MAX_BUCKET);
    if (K != 16 && K != 32) return fail(ctx, -1, "marex_tail_extract_f32: K must be 16 or 32");
    if (((uintptr_t)tails & 15) != 0 || (C % 1) != 0) return fail(ctx, -1, "marex_tail_extract_f32: tails must be 16-byte aligned");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const unsigned ncb = (unsigned)((C + 255) / 256);
    unsigned chunks = (2048 + ncb - 1) / ncb;  // enough workgroups to fill the chip whatever the number of cells
    chunks = chunks < 1 ? 1 : (chunks > 61 ? 61 : chunks);
    const size_t lds = (size_t)(nb + 1) * sizeof(float);
    {
        LaunchTimer lt(ctx, MAREX_K_TAILS);
        if (K == 16)
            hipLaunchKernelGGL(k_tail_extract<16>, dim3(ncb, chunks), dim3(256), lds, ctx->stream, anom, (long)C, doy_start, doy_rows,
                               edges, nb, reinterpret_cast<uint4*>(tails), aux);
        else
            hipLaunchKernelGGL(k_tail_extract<32>, dim3(ncb, chunks), dim3(256), lds, ctx->stream, anom, (long)C, doy_start, doy_rows,
                               edges, nb, reinterpret_cast<uint4*>(tails), aux);
    }
    HIP_TRY(ctx, hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------------------------------------
// K_T from tails: pooled day-of-year histogram quantile (detect.py:2638-2732, 2465-2559).
//
// Workgroup = one tile of TR x TC grid cells (outputs = the inner (TR-2P) x (TC-2P)) and a block of consecutive
// dayofyears, walked in order.  Lane = cell: its private LDS column holds the CUMULATIVE counts of its own wd-day window
// over the levels of a band of 64 bins [B0, B0 + 64): level 0 = everything below the band, level k = bin B0 + k - 1,
// level 65 = everything above (uint16, two levels per dword).  A day's update touches only the keys of the entering and
// the leaving bucket that lie inside the band -- a handful out of the tail, found by walking the sorted chunks until a
// key drops below B0; all other samples of the bucket are one number added to level 0 (aux count minus the keys used).
// A tail that ends inside the band (more samples than keys, last key still >= B0) sends that lane to the bucket's
// anomalies for the keys below its last one: exact for any data, rare by the choice of K.
// Output lanes get pooled cumulative counts by summing the (2P+1)^2 neighbour columns (integers => exact) and walk
// from the previous day's level.
//
// The band FOLLOWS the thresholds: an output whose quantile falls outside the band reports the direction; the tile then
// rebuilds its columns (wd buckets, cheap on tails) around a band further down / up and answers the stragglers, and after
// every day the band is re-centred on the day's range of quantile bins whenever that range comes within MARGIN bins
// of an edge.  Seasonal drift of the thresholds therefore costs a rebuild every few weeks of the walk, not a slower path.
// ------------------------------------------------------------------------------------------------
#define TT_LS 34       // dwords per lane column (68 uint16 levels; stride 34 keeps 8-byte alignment, conflict-free b64)
#define TT_BW 64       // bins per band
#define TT_MARGIN 6    // re-centre when the day's quantile bins come this close to a band edge
#define TT_STEP 56     // band shift when answering stragglers (8 bins of overlap)

struct TailBucket {
    uint4 c0;       // first chunk (8 largest keys)
    unsigned aux;   // count | overflow flag
    int d;          // dayofyear index 0..365
};

template <int P, int TC, int NT, int K, int TR_ = NT / TC>
__global__ void __launch_bounds__(NT, NT == 256 ? 4 : (NT == 512 ? 2 : 1))
k_thr_tails(const uint4* __restrict__ tails, const unsigned short* __restrict__ aux, const float* __restrict__ anom,
            const int* __restrict__ doy_rows, const float* __restrict__ edges, long C, int ny, int nx, int row0, int row1,
            int tiles_x, int Dd, const int* __restrict__ doy_start, const float* __restrict__ centres, int nb, double q, int wd,
            float lower_bound, float upper_bound, float* __restrict__ thr, marex_thr_stats* __restrict__ stats,
            unsigned long long* __restrict__ dbg) {
    constexpr int TR = TR_;
    constexpr int NCELL = TR * TC;
    constexpr int KC = K / 8;
    static_assert(NCELL <= NT && NT - NCELL < 64, "tile does not match the thread count");
    constexpr int OR = TR - 2 * P, OC = TC - 2 * P;
    __shared__ unsigned lev[NT * TT_LS];
    __shared__ unsigned tot_s[NT];
    __shared__ int s_lo[2], s_hi[2];        // per pass (parity): some output's quantile lies below / above the band
    __shared__ int s_iumin[2], s_iumax[2];  // per day (parity): range of the quantile bins found
    __shared__ int s_est_min, s_est_max;

    const int t = threadIdx.x;
    const bool spare = NCELL < NT && t >= NCELL;
    const int tc = t % TC;
    const int tr = (TR > 1) ? (t / TC + P) % TR : 0;  // rows rotated: output rows fill the first waves
    const int ci = spare ? t : tr * TC + tc;
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
        cell_valid = icol < nx && !spare;
        cell = icol < nx ? icol : nx - 1;
    }
    if (!cell_valid) cell = 0;  // valid memory, uniform control flow; contributions are masked
    const bool is_out = !spare && tr >= P && tr < TR - P && tc >= P && tc < TC - P && j < row1 && icol < nx && (ny > 0 ? j >= row0 : true);
    const int d_begin = (int)blockIdx.y * Dd;
    const int ndays = (NDOY - d_begin) < Dd ? (NDOY - d_begin) : Dd;
    const int pd = wd / 2;
    bool land = true;
    if (is_out) land = !(anom[cell] == anom[cell]);  // first kept anomaly row (detect.py:2704)
    if (!__syncthreads_or(is_out && !land)) {
        if (is_out)
            for (int dd = 0; dd < ndays; ++dd) thr[(size_t)(d_begin + dd) * C + cell] = nan_f();
        return;
    }
    if (is_out && land)
        for (int dd = 0; dd < ndays; ++dd) thr[(size_t)(d_begin + dd) * C + cell] = nan_f();

    unsigned* mycol = &lev[ci * TT_LS];
    uint2* mycol2 = reinterpret_cast<uint2*>(mycol);
    typedef __attribute__((address_space(3))) unsigned lds_u32;
    const unsigned col_lds = (unsigned)(size_t)(lds_u32*)mycol;
    const float inv_width = (float)(nb - 1) / (edges[nb] - edges[1]);
    const int base_max = nb - TT_BW > 0 ? nb - TT_BW : 0;
    auto clamp_base = [&](int b) { return b < 0 ? 0 : (b > base_max ? base_max : b); };
    int B0 = 0;
    int BW = nb < TT_BW ? nb : TT_BW;  // levels 1..BW are bins B0 .. B0+BW-1
    constexpr int NLP = (TT_BW + 2 + 1) / 2;  // dwords holding levels 0 .. BW+1

    auto load_bucket = [&](int d0) {
        TailBucket b;
        b.d = d0;
        b.c0 = tails[((size_t)d0 * KC) * C + cell];
        b.aux = aux[(size_t)d0 * C + cell];
        return b;
    };
    // +-1 on level k >= 1 of the lane's packed column
    auto bump = [&](int k, int sgn) {
        const int odd = k & 1, even = k & ~1;
        unsigned addr;
        asm("v_lshl_add_u32 %0, %1, 1, %2" : "=v"(addr) : "v"(even), "v"(col_lds));
        const int v = sgn > 0 ? (odd ? 65536 : 1) : (odd ? -65536 : -1);
        __hip_atomic_fetch_add((lds_u32*)(size_t)addr, (unsigned)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    };
    int mytot = 0;  // valid samples in this lane's window
    unsigned long long n_slow = 0;
    // add (sgn > 0) or remove one bucket
    auto apply_bucket = [&](const TailBucket& b, int sgn) {
        const int cnt = cell_valid ? (int)(b.aux & 0x3FFu) : 0;
        int n_in = 0;
        uint4 ch = b.c0;
        bool more = cell_valid && cnt > 0;
        bool exhausted = false;
        unsigned lastkey = 0;
        for (int jj = 0;; ++jj) {
            const unsigned w[4] = {ch.x, ch.y, ch.z, ch.w};
            bool go = more;
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const unsigned key = (w[u >> 1] >> ((u & 1) * 16)) & 0xFFFFu;
                int lvl = (int)(key >> TAIL_POS_BITS) - B0;  // bin + 1 - B0; an empty key gives <= 0
                lvl = lvl > BW + 1 ? BW + 1 : lvl;
                const bool in = go && lvl >= 1;
                if (__builtin_amdgcn_ballot_w64(in) == 0) {  // sorted: nothing further down is inside the band
                    go = false;
                    break;
                }
                if (in) {
                    bump(lvl, sgn);
                    ++n_in;
                }
                go = in;
                lastkey = key;
            }
            // `go`: this lane's 8th key was still inside the band
            const bool cont = go && jj + 1 < KC;
            if (go && jj + 1 >= KC) exhausted = true;
            if (__builtin_amdgcn_ballot_w64(cont) == 0) break;
            if (cont) ch = tails[((size_t)b.d * KC + jj + 1) * C + cell];
            more = cont;
        }
        // the tail ended inside the band and the bucket holds more samples than keys: the rest from the anomalies
        const bool slow = exhausted && cnt > K;
        if (__builtin_amdgcn_ballot_w64(slow) != 0) {
            const int s0 = doy_start[b.d], nd = doy_start[b.d + 1] - s0;
            for (int pos = 0; pos < nd; ++pos) {
                if (slow) {
                    const float v = anom[(size_t)doy_rows[s0 + pos] * C + cell];
                    const int bin = digitize_bin(v, edges, nb, inv_width);
                    const unsigned key = bin < nb ? tail_key(bin, pos) : 0u;
                    if (key != 0u && key < lastkey) {
                        int lvl = bin + 1 - B0;
                        lvl = lvl > BW + 1 ? BW + 1 : lvl;
                        if (lvl >= 1) {
                            bump(lvl, sgn);
                            ++n_in;
                        }
                    }
                }
            }
            if (slow) ++n_slow;
        }
        const int below = cnt - n_in;  // everything under the band
        if (below > 0) __hip_atomic_fetch_add((lds_u32*)(size_t)col_lds, (unsigned)(sgn > 0 ? below : -below), __ATOMIC_RELAXED,
                                              __HIP_MEMORY_SCOPE_WORKGROUP);
        mytot += sgn > 0 ? cnt : -cnt;
    };
    auto prefix = [&]() -> unsigned {
        unsigned run = 0;
#pragma unroll 4
        for (int i = 0; i < (NLP + 1) / 2; ++i) {
            const uint2 w = mycol2[i];
            const unsigned a0 = (w.x & 0xFFFFu) + run, a1 = (w.x >> 16) + a0;
            const unsigned a2 = (w.y & 0xFFFFu) + a1, a3 = (w.y >> 16) + a2;
            run = a3;
            mycol2[i] = make_uint2(a0 | (a1 << 16), a2 | (a3 << 16));
        }
        return run;
    };
    auto unprefix = [&]() {
        unsigned prev = 0;
#pragma unroll 4
        for (int i = 0; i < (NLP + 1) / 2; ++i) {
            const uint2 w = mycol2[i];
            const unsigned a0 = w.x & 0xFFFFu, a1 = w.x >> 16, a2 = w.y & 0xFFFFu, a3 = w.y >> 16;
            mycol2[i] = make_uint2((a0 - prev) | ((a1 - a0) << 16), (a2 - a1) | ((a3 - a2) << 16));
            prev = a3;
        }
    };
    auto wrap = [&](int d) { return ((d % NDOY) + NDOY) % NDOY; };
    // columns of day d from scratch for the current band
    auto rebuild = [&](int d) {
        for (int r = 0; r < TT_LS / 2; ++r) mycol2[r] = make_uint2(0u, 0u);
        mytot = 0;
        TailBucket cur = load_bucket(wrap(d - pd));
        for (int o = -pd + 1; o <= pd; ++o) {
            const TailBucket nxt = load_bucket(wrap(d + o));
            apply_bucket(cur, +1);
            cur = nxt;
        }
        apply_bucket(cur, +1);
    };
    auto pooled = [&](int k) {
        const unsigned* base = mycol + (k >> 1);
        const int sh16 = (k & 1) * 16;
        int sum = 0;
#pragma unroll
        for (int dr = -P; dr <= P; ++dr)
#pragma unroll
            for (int dc = -P; dc <= P; ++dc) sum += (int)((base[(dr * TC + dc) * TT_LS] >> sh16) & 0xFFFFu);
        return sum;
    };
    auto pooled_tot = [&]() {
        const unsigned* base = &tot_s[tr * TC + tc];
        int sum = 0;
#pragma unroll
        for (int dr = -P; dr <= P; ++dr)
#pragma unroll
            for (int dc = -P; dc <= P; ++dc) sum += (int)base[dr * TC + dc];
        return sum;
    };
    auto window = [&](int start, int (&Wv)[8]) {
        const unsigned* base = mycol + (start >> 1);
        unsigned a0 = 0, a1 = 0, a2 = 0, a3 = 0;
#pragma unroll
        for (int dr = -P; dr <= P; ++dr)
#pragma unroll
            for (int dc = -P; dc <= P; ++dc) {
                const uint2* p2 = reinterpret_cast<const uint2*>(base + (dr * TC + dc) * TT_LS);
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
    // smallest level k < khi whose pooled cumulative count exceeds qpos (khi if none, 0 if level 0 already does);
    // ck = that count, cb = the count one level below
    auto find_level = [&](int hint, int khi, double qpos, int& ck, int& cb) {
        const int qf = (int)floor(qpos);
        int top = (khi - 1) & ~3;
        if (top > 60) top = 60;
        if (top < 0) top = 0;
        int start = ((hint >= 0 ? hint : (khi >> 1)) - 2) & ~3;
        start = start < 0 ? 0 : (start > top ? top : start);
        ck = 0;
        cb = 0;
        for (;;) {
            int Wv[8];
            window(start, Wv);
            const int m = (khi - start) < 8 ? (khi - start) : 8;
            int n = 0;
#pragma unroll
            for (int i = 0; i < 8; ++i) n += (i < m) && (Wv[i] <= qf);
            if (n == 0) {
                if (start == 0) {
                    ck = Wv[0];
                    return 0;
                }
                start -= 4;
                continue;
            }
            if (n == m) {
                if (m < 8 || start >= top) {
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

    // ---- first band of the block: every cell's own bucket of the first day gives an estimate of its quantile bin
    // (the key of rank ceil((1 - q) n) among its n samples); the band is centred on the tile's range of estimates.
    // A bad estimate only costs passes below, never a wrong result.
    if (t == 0) {
        s_est_min = 0x7fffffff;
        s_est_max = -1;
        s_lo[0] = s_lo[1] = s_hi[0] = s_hi[1] = 0;
        s_iumin[0] = s_iumin[1] = 0x7fffffff;
        s_iumax[0] = s_iumax[1] = -1;
    }
    __syncthreads();
    {
        const TailBucket b = load_bucket(d_begin);
        const int cnt = cell_valid ? (int)(b.aux & 0x3FFu) : 0;
        if (cnt > 0) {
            int rank = (int)ceil((1.0 - q) * (double)cnt);
            rank = rank < 1 ? 1 : (rank > 8 ? 8 : rank);
            const unsigned w[4] = {b.c0.x, b.c0.y, b.c0.z, b.c0.w};
            unsigned key = 0;
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (u == rank - 1) key = (w[u >> 1] >> ((u & 1) * 16)) & 0xFFFFu;
            if (key != 0u) {
                const int est = (int)(key >> TAIL_POS_BITS) - 1;
                atomicMin(&s_est_min, est);
                atomicMax(&s_est_max, est);
            }
        }
    }
    __syncthreads();
    {
        const int emin = s_est_min, emax = s_est_max;
        if (emax >= 0) {
            const int span = emax - emin + 1;
            B0 = clamp_base(span <= TT_BW - 2 * TT_MARGIN ? emin - (TT_BW - span) / 2 : emin - TT_MARGIN);
        }
    }
    BW = nb - B0 < TT_BW ? nb - B0 : TT_BW;

    int hint = -1;
    bool need_rebuild = true;
    unsigned long long n_rebuild = 0, n_pass = 0;
    TailBucket pin, pout;
    pin.d = pout.d = 0;
    pin.aux = pout.aux = 0;
    pin.c0 = pout.c0 = make_uint4(0, 0, 0, 0);
    for (int dd = 0; dd < ndays; ++dd) {
        const int d = d_begin + dd;
        const int dpar = dd & 1;
        // ---------------- P1: this lane's column for day d
        if (need_rebuild) {
            rebuild(d);
            need_rebuild = false;
            ++n_rebuild;
            hint = -1;
        } else {
            unprefix();
            apply_bucket(pin, +1);
            apply_bucket(pout, -1);
        }
        tot_s[ci] = prefix();
        if (dd + 1 < ndays) {
            pin = load_bucket(wrap(d + 1 + pd));
            pout = load_bucket(wrap(d - pd));
        }
        // ---------------- P2: quantile level of every output cell; stragglers move the band
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
            if (!resolved) {
                const int tot = pooled_tot();
                if (tot > 0) {
                    const double qpos = q * (double)tot;
                    int ck, cb;
                    int k = find_level(hint, BW + 1, qpos, ck, cb);
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
                        atomicMin(&s_iumin[dpar], iu);
                        atomicMax(&s_iumax[dpar], iu);
                        resolved = true;
                    }
                } else {
                    thr[(size_t)d * C + cell] = nan_f();  // empty window
                    resolved = true;
                }
            }
            __syncthreads();
            const int lo = s_lo[par], hi = s_hi[par];
            if (!lo && !hi) break;
            // stragglers: a band further down (first) or further up than anything tried for this day
            if (lo) {
                B0 = clamp_base(tried_lo - TT_STEP);
                tried_lo = B0;
            } else {
                B0 = clamp_base(tried_hi + TT_STEP);
                tried_hi = B0;
            }
            BW = nb - B0 < TT_BW ? nb - B0 : TT_BW;
            excursion = true;
            rebuild(d);
            tot_s[ci] = prefix();  // same totals (band independent); columns are read after the barrier above
            hint = -1;
            ++n_pass;
        }
        // ---------------- band of the next day
        if (dd + 1 < ndays) {
            const int imin = s_iumin[dpar], imax = s_iumax[dpar];
            if (imax >= 0) {
                const int span = imax - imin + 1;
                const bool near_edge = imin - B0 < TT_MARGIN || (B0 + BW - 1) - imax < TT_MARGIN;
                if (excursion || near_edge) {
                    const int want = clamp_base(span <= TT_BW - 2 * TT_MARGIN ? imin - (TT_BW - span) / 2 : imin - TT_MARGIN);
                    if (want != B0) {
                        B0 = want;
                        BW = nb - B0 < TT_BW ? nb - B0 : TT_BW;
                        need_rebuild = true;
                    } else if (excursion) {
                        need_rebuild = false;  // the last pass already sits on the wanted band
                    }
                }
            }
        }
    }
    // statistics: wave reduction, one set of global atomics per wave
    for (int sft = 32; sft > 0; sft >>= 1) {
        const unsigned a = __shfl_down(kmin, sft, 64), b = __shfl_down(kmax, sft, 64);
        kmin = a < kmin ? a : kmin;
        kmax = b > kmax ? b : kmax;
        nlow += __shfl_down(nlow, sft, 64);
        nhigh += __shfl_down(nhigh, sft, 64);
        n_slow += __shfl_down(n_slow, sft, 64);
    }
    if ((t & 63) == 0) {
        if (kmin != 0xFFFFFFFFu) atomicMin(&stats->min_key, kmin);
        if (kmax != 0u) atomicMax(&stats->max_key, kmax);
        if (nlow) atomicAdd(&stats->n_too_low, nlow);
        if (nhigh) atomicAdd(&stats->n_too_high, nhigh);
        if (dbg && n_slow) atomicAdd(&dbg[1], n_slow);
    }
    if (dbg && t == 0) {
        atomicAdd(&dbg[0], n_rebuild);
        atomicAdd(&dbg[2], n_pass);
        atomicAdd(&dbg[3], (unsigned long long)ndays);
    }
}

extern "C" int marex_hobday_thresholds_tails_f32(marex_ctx* ctx, const void* tails, const uint16_t* aux, int K, const float* anom,
                                                 int64_t T_out, int64_t C, int ny, int nx, const int32_t* doy_start,
                                                 const int32_t* doy_rows, int max_bucket, const float* edges, const float* centres,
                                                 int nb, double q, int wd, int ws, float lower_bound, float upper_bound, int row0,
                                                 int row1, float* thr_doy_major, marex_thr_stats* stats) {
    if (!ctx) return -1;
    if (!tails || !aux || !anom || !doy_start || !doy_rows || !edges || !centres || !thr_doy_major || !stats || T_out <= 0 || C <= 0)
        return fail(ctx, -1, "marex_hobday_thresholds_tails_f32: null pointer or empty shape");
    if (wd < 3 || wd > 365 || (wd & 1) == 0)
        return fail(ctx, -1, "marex_hobday_thresholds_tails_f32: window_days_hobday must be odd and in 3..365");
    if (ws < 1 || (ws & 1) == 0) return fail(ctx, -1, "marex_hobday_thresholds_tails_f32: window_spatial_hobday must be odd");
    if (!(q > 0.0 && q <= 1.0)) return fail(ctx, -1, "marex_hobday_thresholds_tails_f32: q must be in (0, 1]");
    if (K != 16 && K != 32) return fail(ctx, -1, "marex_hobday_thresholds_tails_f32: K must be 16 or 32");
    if (ny == 0) {
        if (ws > 1) return fail(ctx, -1, "marex_hobday_thresholds_tails_f32: spatial pooling needs a structured grid");
        nx = (int)C;
        row0 = 0;
        row1 = 1;
    } else if ((int64_t)ny * nx != C) {
        return fail(ctx, -1, "marex_hobday_thresholds_tails_f32: ny*nx != C");
    } else if (row0 < 0 || row1 > ny || row0 >= row1) {
        return fail(ctx, -1, "marex_hobday_thresholds_tails_f32: need 0 <= row0 < row1 <= ny");
    }
    const int p = ws / 2;
    if (nb < 4 || nb > TAIL_MAX_NB || max_bucket < 1 || max_bucket > TAIL_MAX_BUCKET || p > 3 ||
        (int64_t)max_bucket * wd * ws * ws > 65535)
        return fail(ctx, -4, "marex_hobday_thresholds_tails_f32: shape outside the tail kernel (nb <= %d, buckets <= %d rows, "
                             "ws <= 7, pooled window <= 65535 samples)", TAIL_MAX_NB, TAIL_MAX_BUCKET);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const int tile_pref = ctx_opt(ctx, "THR_TILE", (ny > 0 && p > 0 && max_bucket >= 24) ? 32 : 16);
    const bool big = (ny > 0 && p > 0) && tile_pref == 32 && (row1 - row0) >= 16 && nx >= 16;
    const int NT = big ? 1024 : 256;
    int TR = (ny > 0 && p > 0) ? (big ? 32 : 16) : 1, TC = NT / TR;
    bool tall = false;
    if (big && p == 2 && ctx_opt(ctx, "THR_TALL", 1)) {
        auto ntiles = [&](int tr, int tc) {
            return (long)((nx + tc - 2 * p - 1) / (tc - 2 * p)) * ((row1 - row0 + tr - 2 * p - 1) / (tr - 2 * p));
        };
        tall = ntiles(34, 30) < ntiles(32, 32) && (row1 - row0) >= 30;
        if (tall) TR = 34, TC = 30;
    }
    const int OR = TR - 2 * p, OC = TC - 2 * p;
    const int tiles_x = (nx + OC - 1) / OC, tiles_y = (row1 - row0 + OR - 1) / OR;
    int Dd = ctx_opt(ctx, "THR_DD", 0);
    if (Dd < 1 || Dd > NDOY) {
        // every block pays a window build-up of wd buckets; pick the block length with the least total work per CU slot
        const long cus = device_cus(ctx), tiles = (long)tiles_x * tiles_y;
        const long slots = cus * (big ? 1 : 4);
        long best = -1;
        for (int d = 16; d <= NDOY; ++d) {
            const long blocks = tiles * ((NDOY + d - 1) / d);
            const long cost = ((blocks + slots - 1) / slots) * (d + wd);
            if (best < 0 || cost < best) best = cost, Dd = d;
        }
    }
    dim3 grid((unsigned)(tiles_x * tiles_y), (unsigned)((NDOY + Dd - 1) / Dd));
    unsigned long long* dbg = ctx_debug_counters(ctx);
    const uint4* tl = reinterpret_cast<const uint4*>(tails);
#define MAREX_TT_ARGS tl, aux, anom, doy_rows, edges, (long)C, ny, nx, row0, row1, tiles_x, Dd, doy_start, centres, nb, q, wd, lower_bound, upper_bound, thr_doy_major, stats, dbg
#define MAREX_TT_LAUNCH(PP, TCC, NTT, ...)                                                                                         \
    do {                                                                                                                           \
        if (K == 16)                                                                                                               \
            hipLaunchKernelGGL((k_thr_tails<PP, TCC, NTT, 16, ##__VA_ARGS__>), grid, dim3(NTT), 0, ctx->stream, MAREX_TT_ARGS);    \
        else                                                                                                                       \
            hipLaunchKernelGGL((k_thr_tails<PP, TCC, NTT, 32, ##__VA_ARGS__>), grid, dim3(NTT), 0, ctx->stream, MAREX_TT_ARGS);    \
    } while (0)
    {
        LaunchTimer lt(ctx, MAREX_K_THRESHOLDS);
        if (TR == 1)
            MAREX_TT_LAUNCH(0, 256, 256);
        else if (big && p == 1)
            MAREX_TT_LAUNCH(1, 32, 1024);
        else if (big && p == 2 && tall)
            MAREX_TT_LAUNCH(2, 30, 1024, 34);
        else if (big && p == 2)
            MAREX_TT_LAUNCH(2, 32, 1024);
        else if (big)
            MAREX_TT_LAUNCH(3, 32, 1024);
        else if (p == 1)
            MAREX_TT_LAUNCH(1, 16, 256);
        else if (p == 2)
            MAREX_TT_LAUNCH(2, 16, 256);
        else
            MAREX_TT_LAUNCH(3, 16, 256);
    }
#undef MAREX_TT_LAUNCH
#undef MAREX_TT_ARGS
    HIP_TRY(ctx, hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------------------------------------
// K_M from tails: extreme[t, c] = anom[t, c] >= thr[doy(t), c]  (detect.py:2003-2004) without reading the anomalies.
// Every sample at or above the threshold sits at the top of its bucket's tail: keys whose bin lies above the bin of
// the threshold are extremes, keys in the threshold's own bin are compared as numbers (a handful per wave and day),
// the first key below ends the walk.  Buckets whose tail does not reach below the threshold, and buckets holding
// values beyond the edge table, are decided on the anomalies themselves (float4 rows, like the plain mask kernel).
// Lane = 4 consecutive cells: 16-byte key chunks of 4 cells are one contiguous 64-byte run, mask stores are 4 bytes.
// ------------------------------------------------------------------------------------------------
template <int K>
__global__ void __launch_bounds__(256)
k_mask_tails(const uint4* __restrict__ tails, const unsigned short* __restrict__ aux, const float* __restrict__ anom,
             const float* __restrict__ edges, int nb, const float* __restrict__ thr, const int* __restrict__ doy_start,
             const int* __restrict__ doy_rows, long C, long c0, long c1, unsigned char* __restrict__ out,
             unsigned long long* __restrict__ n_true, unsigned long long* __restrict__ dbg) {
    constexpr int KC = K / 8;
    const int nchunk = (int)gridDim.y;
    const int dA = (int)blockIdx.y * NDOY / nchunk, dB = ((int)blockIdx.y + 1) * NDOY / nchunk;
    const long c = c0 + ((long)blockIdx.x * 256 + threadIdx.x) * 4;
    unsigned cnt_true = 0;
    unsigned long long n_slow = 0;
    if (c < c1) {
        const float inv_width = (float)(nb - 1) / (edges[nb] - edges[1]);
        for (int d = dA; d < dB; ++d) {
            const int r0 = doy_start[d], nd = doy_start[d + 1] - r0;
            if (nd == 0) continue;
            const float4 th4 = *reinterpret_cast<const float4*>(thr + (size_t)d * C + c);
            const float tv[4] = {th4.x, th4.y, th4.z, th4.w};
            const uint2 ax = *reinterpret_cast<const uint2*>(aux + (size_t)d * C + c);
            const unsigned av[4] = {ax.x & 0xFFFFu, ax.x >> 16, ax.y & 0xFFFFu, ax.y >> 16};
            unsigned bits[4][4];
            int kt[4];
            bool slow = false, open[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                bits[i][0] = bits[i][1] = bits[i][2] = bits[i][3] = 0u;
                const bool isnum = tv[i] == tv[i];
                kt[i] = isnum ? digitize_bin(tv[i], edges, nb, inv_width) : 0x7fff;  // NaN threshold: nothing is extreme
                slow = slow || (isnum && (av[i] & 0x8000u));                            // values beyond the table: look at them
                open[i] = isnum && (av[i] & 0x3FFu) > 0;                                // still walking this cell's tail
            }
            for (int jj = 0; jj < KC; ++jj) {
                bool any_open = false;
#pragma unroll
                for (int i = 0; i < 4; ++i) any_open = any_open || open[i];
                if (__builtin_amdgcn_ballot_w64(any_open && !slow) == 0) break;
                uint4 ch[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) ch[i] = tails[((size_t)d * KC + jj) * C + c + i];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const unsigned w[4] = {ch[i].x, ch[i].y, ch[i].z, ch[i].w};
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const unsigned key = (w[u >> 1] >> ((u & 1) * 16)) & 0xFFFFu;
                        const int bin = (int)(key >> TAIL_POS_BITS) - 1;  // -1 for an empty key
                        const int pos = (int)(key & (TAIL_MAX_BUCKET - 1));
                        if (open[i]) {
                            bool ext = bin > kt[i];
                            if (bin == kt[i])  // the threshold's own bin: compare the numbers
                                ext = anom[(size_t)doy_rows[r0 + pos] * C + c + i] >= tv[i];
                            if (bin < kt[i]) open[i] = false;  // sorted: nothing further down reaches the threshold
                            if (ext) {
#pragma unroll
                                for (int wi = 0; wi < 4; ++wi)
                                    if ((pos >> 5) == wi) bits[i][wi] |= 1u << (pos & 31);
                            }
                        }
                    }
                    // tail used up while still at or above the threshold's bin, and the bucket holds more samples than keys
                    if (open[i] && jj == KC - 1 && (int)(av[i] & 0x3FFu) > K) slow = true;
                }
            }
            if (slow) {
                ++n_slow;
                for (int r = 0; r < nd; ++r) {
                    const size_t off = (size_t)doy_rows[r0 + r] * C + c;
                    const float4 a = *reinterpret_cast<const float4*>(anom + off);
                    const unsigned m0 = a.x >= tv[0], m1 = a.y >= tv[1], m2 = a.z >= tv[2], m3 = a.w >= tv[3];
                    cnt_true += m0 + m1 + m2 + m3;
                    __builtin_nontemporal_store(m0 | (m1 << 8) | (m2 << 16) | (m3 << 24), reinterpret_cast<unsigned*>(out + off));
                }
            } else {
#pragma unroll
                for (int wi = 0; wi < 4; ++wi) {
                    const unsigned b0 = bits[0][wi], b1 = bits[1][wi], b2 = bits[2][wi], b3 = bits[3][wi];
                    cnt_true += __popc(b0) + __popc(b1) + __popc(b2) + __popc(b3);
                    const int rlo = wi * 32;
                    if (rlo >= nd) break;
                    const int rn = (nd - rlo) < 32 ? (nd - rlo) : 32;
                    for (int rr = 0; rr < rn; ++rr) {
                        const size_t off = (size_t)doy_rows[r0 + rlo + rr] * C + c;
                        const unsigned m = ((b0 >> rr) & 1u) | (((b1 >> rr) & 1u) << 8) | (((b2 >> rr) & 1u) << 16) | (((b3 >> rr) & 1u) << 24);
                        __builtin_nontemporal_store(m, reinterpret_cast<unsigned*>(out + off));
                    }
                }
            }
        }
    }
    if (n_true) {
        for (int s = 32; s > 0; s >>= 1) cnt_true += __shfl_down(cnt_true, s, 64);
        if ((threadIdx.x & 63) == 0 && cnt_true) atomicAdd(n_true, (unsigned long long)cnt_true);
    }
    if (dbg) {
        for (int s = 32; s > 0; s >>= 1) n_slow += __shfl_down(n_slow, s, 64);
        if ((threadIdx.x & 63) == 0 && n_slow) atomicAdd(&dbg[4], n_slow);
    }
}

extern "C" int marex_mask_ge_doy_tails_f32(marex_ctx* ctx, const void* tails, const uint16_t* aux, int K, const float* anom,
                                           const float* edges, int nb, const float* thr_doy_major, const int32_t* doy_start,
                                           const int32_t* doy_rows, int64_t T_out, int64_t C, int64_t c0, int64_t c1,
                                           uint8_t* extreme, unsigned long long* n_true) {
    if (!ctx) return -1;
    if (!tails || !aux || !anom || !edges || !thr_doy_major || !doy_start || !doy_rows || !extreme || T_out <= 0 || C <= 0 || nb < 4)
        return fail(ctx, -1, "marex_mask_ge_doy_tails_f32: null pointer or empty shape");
    if (c0 < 0 || c1 > C || c0 >= c1) return fail(ctx, -1, "marex_mask_ge_doy_tails_f32: need 0 <= c0 < c1 <= C");
    if (K != 16 && K != 32) return fail(ctx, -1, "marex_mask_ge_doy_tails_f32: K must be 16 or 32");
    const bool vec = (C % 4 == 0) && (c0 % 4 == 0) && (c1 % 4 == 0) && (((uintptr_t)anom | (uintptr_t)thr_doy_major) % 16 == 0) &&
                     ((uintptr_t)extreme % 4 == 0) && ((uintptr_t)aux % 8 == 0) && nb < 0x7fff;
    if (!vec)  // shapes the 4-cell kernel does not cover: the plain compare on the anomalies
        return marex_mask_ge_doy_f32(ctx, anom, thr_doy_major, doy_start, doy_rows, T_out, C, c0, c1, extreme, n_true);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    {
        LaunchTimer lt(ctx, MAREX_K_MASK);
        const unsigned ncb = (unsigned)(((c1 - c0) / 4 + 255) / 256);
        unsigned chunks = (4096 + ncb - 1) / ncb;
        chunks = chunks < MASK_DOY_CHUNKS ? MASK_DOY_CHUNKS : (chunks > 61 ? 61 : chunks);
        unsigned long long* dbg = ctx_debug_counters(ctx);
        const uint4* tl = reinterpret_cast<const uint4*>(tails);
        if (K == 16)
            hipLaunchKernelGGL(k_mask_tails<16>, dim3(ncb, chunks), dim3(256), 0, ctx->stream, tl, aux, anom, edges, nb, thr_doy_major,
                               doy_start, doy_rows, (long)C, (long)c0, (long)c1, extreme, n_true, dbg);
        else
            hipLaunchKernelGGL(k_mask_tails<32>, dim3(ncb, chunks), dim3(256), 0, ctx->stream, tl, aux, anom, edges, nb, thr_doy_major,
                               doy_start, doy_rows, (long)C, (long)c0, (long)c1, extreme, n_true, dbg);
    }
    HIP_TRY(ctx, hipGetLastError());
    return 0;
}
