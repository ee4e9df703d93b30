// marex_tails.hip -- K_X: tail extraction, K_T: day-of-year thresholds from tails, K_M: extreme mask from tails
#include "marex_common.hip.h"
#include "marex_tails.hip.h"

// ------------------------------------------------------------------------------------------------
// device helpers shared by the three kernels
// ------------------------------------------------------------------------------------------------
typedef __amdgpu_buffer_rsrc_t tl_rsrc_t;
__device__ __forceinline__ tl_rsrc_t tl_make_rsrc(const void* base) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, -1, 0x00020000);
}
// one 16-byte chunk: wave-uniform base (SGPRs), 32-bit lane byte offset, 32-bit uniform byte offset
__device__ __forceinline__ uint4 tl_load_chunk(tl_rsrc_t r, unsigned voff, unsigned soff) {
    typedef unsigned u4 __attribute__((ext_vector_type(4)));
    const u4 v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, (int)soff, 0);
    return make_uint4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ unsigned pk_sub_sat_u16(unsigned a, unsigned b) {
    unsigned r;
    asm("v_pk_sub_u16 %0, %1, %2 clamp" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ unsigned pk_add_u16(unsigned a, unsigned b) {
    unsigned r;
    asm("v_pk_add_u16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ unsigned pk_sub_u16(unsigned a, unsigned b) {
    unsigned r;
    asm("v_pk_sub_u16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
// min(x, 1) per half through the instruction itself: the compiler rewrites the generic vector min with the constant 1 into
// (x != 0) per 16-bit element -- two compares, two selects and a permute per register instead of one v_pk_min_u16
__device__ __forceinline__ unsigned pk_min1_u16(unsigned a) {
    unsigned r;
    asm("v_pk_min_u16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(0x00010001u));
    return r;
}
// number of keys of a chunk that are > lim (lim replicated in both halves of lim_rep); the chunk is sorted descending, so
// these are its FIRST keys.  10 instructions for 8 keys.
__device__ __forceinline__ int tl_count_above(const uint4& ch, unsigned lim_rep) {
    const unsigned m0 = pk_min1_u16(pk_sub_sat_u16(ch.x, lim_rep)), m1 = pk_min1_u16(pk_sub_sat_u16(ch.y, lim_rep));
    const unsigned m2 = pk_min1_u16(pk_sub_sat_u16(ch.z, lim_rep)), m3 = pk_min1_u16(pk_sub_sat_u16(ch.w, lim_rep));
    const unsigned s = (m0 + m1) + (m2 + m3);
    return (int)((s & 0xFFFFu) + (s >> 16));
}
// key u (0..7) of a chunk; u is a compile-time constant at every call
__device__ __forceinline__ unsigned tl_key(const uint4& ch, int u) {
    const unsigned w = u < 2 ? ch.x : (u < 4 ? ch.y : (u < 6 ? ch.z : ch.w));
    return (u & 1) ? (w >> 16) : (w & 0xFFFFu);
}
// bin + 1 of key u: bits 7..15 of its half
__device__ __forceinline__ int tl_bin1(const uint4& ch, int u) {
    const unsigned w = u < 2 ? ch.x : (u < 4 ? ch.y : (u < 6 ? ch.z : ch.w));
    return (u & 1) ? (int)(w >> (16 + TAIL_POS_BITS)) : (int)__builtin_amdgcn_ubfe(w, TAIL_POS_BITS, 16 - TAIL_POS_BITS);
}

// ------------------------------------------------------------------------------------------------
// K_X: tails of an anomaly field (the counting stage of detect.py:2622-2648 in the form the threshold and mask kernels
// consume, see marex_tails.hip.h).  Thread = cell, two neighbouring dayofyears at a time as packed pairs: 16 rows of
// both buckets in flight, np.digitize, keys, a 16-key sorting network per batch and a bitonic merge of two batches into
// one 32-key list.  Reads every anomaly once: coalesced 256-byte row segments per wave, like the mask kernel.
// The kernel is bound by instruction issue, scalar instructions included: row addresses come from a table of 64-bit
// row offsets (one scalar load + add per row instead of a 64-bit multiply chain), full batches run without predicates.
// ------------------------------------------------------------------------------------------------
__global__ void k_row_offsets(const int* __restrict__ doy_rows, long T_out, long C, long long* __restrict__ off, int elem_bytes) {
    const long r = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (r < T_out) off[r] = (long long)doy_rows[r] * C * (long long)elem_bytes;
}

__global__ void __launch_bounds__(256)
k_tail_extract(const float* __restrict__ anom, long C, const int* __restrict__ doy_start, const long long* __restrict__ row_off,
               const float* __restrict__ edges, int nb, int NPER, int list_rows, uint4* __restrict__ lists,
               unsigned* __restrict__ aux, const int* __restrict__ skip_chunks) {
    extern __shared__ float e[];  // [nb + 1]
    const int nch = list_rows <= 16 ? 2 : 4;  // 16-byte chunks per list
    const int tid = threadIdx.x;
    for (int i = tid; i <= nb; i += 256) e[i] = edges[i];
    __syncthreads();
    const float e_first = e[1], e_delta = e[2] - e[1], e_last = e[nb];
    const float inv_width = (float)(nb - 1) / (e_last - e_first);
    // arange table (NumPy float32 arange, contract C4) whose fused guess is provably within one bin: the lean digitize of
    // the anomaly kernel; any other increasing table: the general search
    bool lean = edges_are_arange(e, nb);
    {
        const double m = fabs((double)e_first) > fabs((double)e_last) ? fabs((double)e_first) : fabs((double)e_last);
        lean = lean && e_delta > 0.f && ((double)nb + 2.0 * m / (double)e_delta) * (1.0 / 1048576.0) < 1.0 / 256.0;
    }
    const float c0 = (1.0f - e_first * inv_width) - 0.0078125f, nbm1f = (float)(nb - 1);
    const long c = (long)blockIdx.x * 256 + tid;
    const bool active = c < C;
    const unsigned lane_off = (unsigned)(active ? c : C - 1) * 4u;  // byte offset of the lane's cell inside a row
    const char* abase = reinterpret_cast<const char*>(anom);
    const int npairs = NDOY / 2;
    const int pA = (int)blockIdx.y * npairs / (int)gridDim.y, pB = ((int)blockIdx.y + 1) * npairs / (int)gridDim.y;
    // the whole walk once per digitize flavour (uniform choice at the top, no branch per sample)
    auto body = [&](auto lean_tag) {
    constexpr bool LEAN = decltype(lean_tag)::value;
    auto digit = [&](float a) -> int {
        if (LEAN) {
            const float f = __builtin_fmaf(a, inv_width, c0);
            const float t = __builtin_amdgcn_fmed3f(__builtin_floorf(f), 0.0f, nbm1f);
            const float ehi = e_first + t * e_delta;  // edges[t + 1], the table's own arithmetic (separately rounded)
            const int k = (int)t + (a >= ehi ? 1 : 0);
            return (a == a) ? k : nb;
        }
        return digitize_bin(a, e, nb, inv_width);
    };
    for (int pp = pA; pp < pB; ++pp) {
        if (skip_chunks && skip_chunks[pp >> 1]) continue;  // this group of 4 dayofyears got its lists from the anomaly kernel
        const int d0 = 2 * pp, d1 = d0 + 1;
        const int s0 = doy_start[d0], n0 = doy_start[d0 + 1] - s0;
        const int s1 = doy_start[d1], n1 = doy_start[d1 + 1] - s1;
        unsigned cnt = 0, ovf = 0;  // packed: low half dayofyear d0, high half d1 (ovf: tail_ovf_add states)
        for (int p = 0; p < NPER; ++p) {
            unsigned half[2][16];
            const int l0 = p * list_rows, l1 = l0 + list_rows;  // rows of this list
            const int e0 = n0 < l1 ? n0 : l1, e1 = n1 < l1 ? n1 : l1;  // ... that exist in either bucket
            const bool short_list = l0 + 16 >= e0 && l0 + 16 >= e1;  // uniform: rows 16.. of the list do not exist
#pragma unroll
            for (int hb = 0; hb < 2; ++hb) {  // two batches of 16 rows
                if (hb == 1 && short_list) break;
                const int r = l0 + hb * 16;
                float va[16], vb[16];
                if (r + 16 <= e0 && r + 16 <= e1) {  // full batch (uniform): no predicates
#pragma unroll
                    for (int u = 0; u < 16; ++u) {
                        va[u] = *reinterpret_cast<const float*>(abase + row_off[s0 + r + u] + lane_off);
                        vb[u] = *reinterpret_cast<const float*>(abase + row_off[s1 + r + u] + lane_off);
                    }
                } else {
#pragma unroll
                    for (int u = 0; u < 16; ++u) {
                        const int pos = r + u;  // uniform
                        va[u] = pos < e0 ? *reinterpret_cast<const float*>(abase + row_off[s0 + pos] + lane_off) : nan_f();
                        vb[u] = pos < e1 ? *reinterpret_cast<const float*>(abase + row_off[s1 + pos] + lane_off) : nan_f();
                    }
                }
#pragma unroll
                for (int u = 0; u < 16; ++u) {
                    const int pos = r + u;
                    const int ba = digit(va[u]), bb = digit(vb[u]);
                    const bool oa = ba < nb, ob = bb < nb;
                    const unsigned ka = oa ? tail_key(ba, pos) : 0u, kb = ob ? tail_key(bb, pos) : 0u;
                    cnt += (oa ? 1u : 0u) + (ob ? 0x10000u : 0u);
                    half[hb][u] = ka | (kb << 16);
                }
                {  // samples beyond the table (false for NaN): next to never -- one test per batch, their positions in a re-walk
                    float mx = -__builtin_inff();
#pragma unroll
                    for (int u = 0; u < 16; ++u) mx = fmaxf(mx, fmaxf(va[u], vb[u]));  // maxNum skips NaN
                    if (__builtin_amdgcn_ballot_w64(mx >= e_last) != 0) {
                        unsigned sa = ovf & 0xFFFFu, sb = ovf >> 16;
#pragma unroll
                        for (int u = 0; u < 16; ++u) {
                            if (va[u] >= e_last) sa = tail_ovf_add(sa, (unsigned)(r + u));
                            if (vb[u] >= e_last) sb = tail_ovf_add(sb, (unsigned)(r + u));
                        }
                        ovf = sa | (sb << 16);
                    }
                }
                sort16_desc(half[hb]);
            }
            // two sorted runs -> one sorted list of 32: first ++ reverse(second) is bitonic
            unsigned v[32];
            if (short_list) {
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    v[i] = half[0][i];
                    v[16 + i] = 0u;
                }
            } else {
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    v[i] = half[0][i];
                    v[16 + i] = half[1][15 - i];
                }
                bitonic_merge_desc<32>(v);
            }
            if (active) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (j >= nch) break;
                    uint4 w0, w1;
                    unsigned* a = reinterpret_cast<unsigned*>(&w0);
                    unsigned* b = reinterpret_cast<unsigned*>(&w1);
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const unsigned x = v[8 * j + 2 * i], y = v[8 * j + 2 * i + 1];
                        a[i] = (x & 0xFFFFu) | (y << 16);
                        b[i] = (x >> 16) | (y & 0xFFFF0000u);
                    }
                    lists[(((size_t)d0 * NPER + p) * nch + j) * C + c] = w0;
                    lists[(((size_t)d1 * NPER + p) * nch + j) * C + c] = w1;
                }
            }
        }
        if (active) {
            aux[(size_t)d0 * C + c] = tail_aux_word(cnt & 0xFFFFu, ovf & 0xFFFFu);
            aux[(size_t)d1 * C + c] = tail_aux_word(cnt >> 16, ovf >> 16);
        }
    }
    };
    if (lean)
        body(std::true_type{});
    else
        body(std::false_type{});
}

static bool tails_geometry(int max_bucket, int list_rows, int& nper, int& nch) {
    if (max_bucket < 1 || max_bucket > TAIL_MAX_BUCKET || list_rows < 8 || list_rows > 32) return false;
    nper = (max_bucket + list_rows - 1) / list_rows;
    nch = list_rows <= 16 ? 2 : 4;
    return true;
}

extern "C" int marex_tail_lists(int max_bucket, int list_rows) {
    int nper, nch;
    return tails_geometry(max_bucket, list_rows, nper, nch) ? nper : -1;
}

int marex_tail_extract_impl(marex_ctx* ctx, const float* anom, int64_t T_out, int64_t C, const int32_t* doy_start,
                            const int32_t* doy_rows, int max_bucket, const float* edges, int nb, int list_rows, void* lists,
                            uint32_t* aux, const int* skip_chunks) {
    if (!ctx) return -1;
    if (!anom || !doy_start || !doy_rows || !edges || !lists || !aux || T_out <= 0 || C <= 0)
        return fail(ctx, -1, "marex_tail_extract_f32: null pointer or empty shape");
    if (nb < 4 || nb > TAIL_MAX_NB) return fail(ctx, -4, "marex_tail_extract_f32: nb must be in 4..%d", TAIL_MAX_NB);
    int nper, nch;
    if (!tails_geometry(max_bucket, list_rows, nper, nch))
        return fail(ctx, -4, "marex_tail_extract_f32: dayofyear buckets must hold 1..%d rows, lists 8..32 rows", TAIL_MAX_BUCKET);
    if (((uintptr_t)lists & 15) != 0) return fail(ctx, -1, "marex_tail_extract_f32: lists must be 16-byte aligned");
    if (C * 4 > 0xFFFFFFFFll) return fail(ctx, -4, "marex_tail_extract_f32: more than 2^30 cells");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const size_t need = (size_t)T_out * sizeof(long long);
    if (need > ctx->row_off_bytes) {
        if (ctx->row_off) HIP_TRY(ctx, hipFree(ctx->row_off));
        ctx->row_off = nullptr;
        ctx->row_off_bytes = 0;
        HIP_TRY(ctx, hipMalloc((void**)&ctx->row_off, need));
        ctx->row_off_bytes = need;
    }
    const unsigned ncb = (unsigned)((C + 255) / 256);
    unsigned chunks = (2048 + ncb - 1) / ncb;  // enough workgroups to fill the chip whatever the number of cells
    chunks = chunks < 1 ? 1 : (chunks > 61 ? 61 : chunks);
    const size_t lds = (size_t)(nb + 1) * sizeof(float);
    {
        LaunchTimer lt(ctx, MAREX_K_TAILS);
        hipLaunchKernelGGL(k_row_offsets, dim3((unsigned)((T_out + 255) / 256)), dim3(256), 0, ctx->stream, doy_rows, (long)T_out, (long)C,
                           ctx->row_off, (int)sizeof(float));
        hipLaunchKernelGGL(k_tail_extract, dim3(ncb, chunks), dim3(256), lds, ctx->stream, anom, (long)C, doy_start, ctx->row_off, edges,
                           nb, nper, list_rows, reinterpret_cast<uint4*>(lists), aux, skip_chunks);
    }
    HIP_TRY(ctx, hipGetLastError());
    return 0;
}

extern "C" int marex_tail_extract_f32(marex_ctx* ctx, const float* anom, int64_t T_out, int64_t C, const int32_t* doy_start,
                                      const int32_t* doy_rows, int max_bucket, const float* edges, int nb, int list_rows,
                                      void* lists, uint32_t* aux) {
    return marex_tail_extract_impl(ctx, anom, T_out, C, doy_start, doy_rows, max_bucket, edges, nb, list_rows, lists, aux, nullptr);
}

// ------------------------------------------------------------------------------------------------
// K_T from tails: pooled day-of-year histogram quantile (detect.py:2638-2732, 2465-2559).
//
// Workgroup = one tile of TR x TC grid cells (outputs = the inner (TR-2P) x (TC-2P)) and a block of consecutive
// dayofyears, walked in order.  Lane = cell: its private LDS column holds the CUMULATIVE counts of its own wd-day window
// over the levels of a band of 64 bins [B0, B0 + 64): level 0 = everything below the band, level k = bin B0 + k - 1,
// level 65 = everything above (uint16, two levels per dword).  A day's update touches only the keys of the entering and
// the leaving bucket that lie inside the band: the first keys of every sorted list (counted with packed 16-bit
// arithmetic, no search); all other samples of the bucket are one number added to level 0 (aux count minus the keys used).
// Output lanes get pooled cumulative counts by summing the (2P+1)^2 neighbour columns (integers => exact) and walk
// from the previous day's level.
//
// The band FOLLOWS the thresholds: an output whose quantile falls outside the band reports the direction; the tile then
// rebuilds its columns (wd buckets, cheap on tails) around a band further down / up and answers the stragglers, and after
// every day the band is re-centred on the day's range of quantile bins whenever that range comes within MARGIN bins
// of an edge.  Seasonal drift of the thresholds therefore costs a rebuild every few weeks of the walk, not a slower path.
//
// The kernel is bound by instruction ISSUE (a CDNA4 SIMD starts about one instruction of any kind per four cycles),
// so the inner pieces are written for instruction count: packed 16-bit prefix sums, counted key prefixes instead of
// per-key branches, buffer addressing with wave-uniform bases.
// ------------------------------------------------------------------------------------------------

// The first n keys of a chunk (n <= 8, all at or above the band's first bin) added to (SGN > 0) or removed from the lane's packed
// level column: level = min(bin + 1 - B0, BW + 1), two uint16 levels per dword.  Lanes drop out of the loop for good as the key
// index passes their n, so the execution mask only ever shrinks: v_cmpx + s_cbranch_execz per key, the mask saved and restored
// once -- hipcc's form of the same loop (compare, branch on the ballot, s_and_saveexec, branch, restore, per key) spent 5 scalar
// and branch instructions per key where this spends one, in the loop that is most of the threshold kernel's column update.
#define MAREX_BUMP_KEY(u, w, ext)                                                                                   \
    "v_cmpx_lt_u32_e32 vcc, " #u ", %[n]\n\ts_cbranch_execz 1f\n\t" ext "\n\t"                                      \
    "v_subrev_u32_e32 %[t], %[b0], %[t]\n\tv_min_i32_e32 %[t], %[hi], %[t]\n\tv_and_b32_e32 %[a], -2, %[t]\n\t"       \
    "v_and_b32_e32 %[t], 1, %[t]\n\tv_lshl_add_u32 %[a], %[a], 1, %[col]\n\t"
#define MAREX_BUMP_LO(w) "v_bfe_u32 %[t], %[" #w "], 7, 9"
#define MAREX_BUMP_HI(w) "v_lshrrev_b32_e32 %[t], 23, %[" #w "]"
template <int SGN>
__device__ __forceinline__ void tl_bump_chunk(const uint4& ch, int n, int B0, int BWp1, unsigned col_lds) {
    unsigned t, a;
    unsigned long long sv;
    const int mul = SGN > 0 ? 65535 : -65535;
#define MAREX_BUMP_ADD(one) "v_mad_i32_i24 %[t], %[t], %[mul], " one "\n\tds_add_u32 %[a], %[t]\n\t"
#define MAREX_BUMP_ALL(one)                                                                                          \
    "s_mov_b64 %[sv], exec\n\t"                                                                                      \
    MAREX_BUMP_KEY(0, x, MAREX_BUMP_LO(x)) MAREX_BUMP_ADD(one) MAREX_BUMP_KEY(1, x, MAREX_BUMP_HI(x)) MAREX_BUMP_ADD(one)   \
    MAREX_BUMP_KEY(2, y, MAREX_BUMP_LO(y)) MAREX_BUMP_ADD(one) MAREX_BUMP_KEY(3, y, MAREX_BUMP_HI(y)) MAREX_BUMP_ADD(one)   \
    MAREX_BUMP_KEY(4, z, MAREX_BUMP_LO(z)) MAREX_BUMP_ADD(one) MAREX_BUMP_KEY(5, z, MAREX_BUMP_HI(z)) MAREX_BUMP_ADD(one)   \
    MAREX_BUMP_KEY(6, w, MAREX_BUMP_LO(w)) MAREX_BUMP_ADD(one) MAREX_BUMP_KEY(7, w, MAREX_BUMP_HI(w)) MAREX_BUMP_ADD(one)   \
    "1:\n\ts_mov_b64 exec, %[sv]"
    if (SGN > 0)
        asm volatile(MAREX_BUMP_ALL("1")
                     : [t] "=&v"(t), [a] "=&v"(a), [sv] "=&s"(sv)
                     : [n] "v"(n), [x] "v"(ch.x), [y] "v"(ch.y), [z] "v"(ch.z), [w] "v"(ch.w), [b0] "s"(B0), [hi] "s"(BWp1), [col] "v"(col_lds), [mul] "s"(mul)
                     : "vcc", "memory");
    else
        asm volatile(MAREX_BUMP_ALL("-1")
                     : [t] "=&v"(t), [a] "=&v"(a), [sv] "=&s"(sv)
                     : [n] "v"(n), [x] "v"(ch.x), [y] "v"(ch.y), [z] "v"(ch.z), [w] "v"(ch.w), [b0] "s"(B0), [hi] "s"(BWp1), [col] "v"(col_lds), [mul] "s"(mul)
                     : "vcc", "memory");
#undef MAREX_BUMP_ALL
#undef MAREX_BUMP_ADD
}
#undef MAREX_BUMP_KEY
#undef MAREX_BUMP_LO
#undef MAREX_BUMP_HI

#define TT_LS 34       // dwords per lane column (68 uint16 levels; stride 34 keeps 8-byte alignment, conflict-free b64)
#ifndef TT_BW
#define TT_BW 64       // bins per band (experiments: -DTT_BW=48 -DTT_STEP=40; must stay <= 64: the column has 68 levels)
#endif
#define TT_MARGIN 6    // re-centre when the day's quantile bins come this close to a band edge
#ifndef TT_STEP
#define TT_STEP 56     // band shift when answering stragglers (8 bins of overlap)
#endif
static_assert(TT_STEP <= TT_BW - 8, "straggler passes must overlap: a quantile bin between two tried bands would never be found");
#define TT_NPF 2       // chunk-0s of a bucket prefetched across the barrier (the other lists are loaded at use)

template <int NPERT>
struct TailBucket {
    uint4 c0[NPERT < TT_NPF ? NPERT : TT_NPF];
    unsigned aux;  // count | overflow flag
    int d;         // dayofyear index 0..365
};

template <int P, int TC, int NT, int NPERT, int TR_ = NT / TC>
__global__ void __launch_bounds__(NT, NT == 256 ? 4 : (NT == 512 ? 2 : 1))
k_thr_tails(const uint4* __restrict__ lists, const unsigned* __restrict__ aux, int NPER, int nch, const float* __restrict__ anom,
            long C, int ny, int nx, int row0, int row1, int tiles_x, int Dd, const float* __restrict__ centres, int nb, double q,
            int wd, float lower_bound, float upper_bound, float* __restrict__ thr, marex_thr_stats* __restrict__ stats,
            unsigned long long* __restrict__ dbg, int pass_limit) {
    constexpr int TR = TR_;
    constexpr int NCELL = TR * TC;
    constexpr int NPF = NPERT < TT_NPF ? NPERT : TT_NPF;
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
    const int base_max = nb - TT_BW > 0 ? nb - TT_BW : 0;
    auto clamp_base = [&](int b) { return b < 0 ? 0 : (b > base_max ? base_max : b); };
    int B0 = 0;
    int BW = nb < TT_BW ? nb : TT_BW;      // levels 1..BW are bins B0 .. B0+BW-1
    constexpr int NLP = (TT_BW + 2 + 1) / 2;  // dwords holding levels 0 .. BW+1
    const unsigned voff = (unsigned)cell * 16u;               // lane byte offset inside one chunk row of C cells
    const unsigned chunk_row = (unsigned)C * 16u;              // bytes of one chunk row (C cells)
    const size_t day_stride = (size_t)NPER * nch * (size_t)C;  // uint4 elements per dayofyear

    auto load_bucket = [&](int d0) {
        TailBucket<NPERT> b;
        b.d = d0;
        const tl_rsrc_t r = tl_make_rsrc(lists + (size_t)d0 * day_stride);
#pragma unroll
        for (int p = 0; p < NPF; ++p)
            b.c0[p] = (p < NPER) ? tl_load_chunk(r, voff, (unsigned)(nch * p) * chunk_row) : make_uint4(0, 0, 0, 0);
        b.aux = aux[(size_t)d0 * C + cell];
        return b;
    };
    // +-1 on level k >= 1 of the lane's packed column
    auto bump = [&](int k, int sgn) {
        const int odd = k & 1;
        unsigned addr;
        asm("v_lshl_add_u32 %0, %1, 1, %2" : "=v"(addr) : "v"(k - odd), "v"(col_lds));
        int v;
        const int mul = sgn > 0 ? 65535 : -65535, one = sgn > 0 ? 1 : -1;
        asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(v) : "v"(odd), "v"(mul), "v"(one));
        __hip_atomic_fetch_add((lds_u32*)(size_t)addr, (unsigned)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    };
    int mytot = 0;  // valid samples in this lane's window
    // the first n keys of a chunk (n <= 8, all inside the band) into the column
    constexpr bool ctx_asm_bump = true;  // (the C++ loop below is what it replaces; kept for reference builds)
    auto bump_chunk = [&](const uint4& ch, int n, int sgn, int nfix) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const bool on = n > u;
            if (u >= nfix && __builtin_amdgcn_ballot_w64(on) == 0) break;
            if (on) {
                int lvl = tl_bin1(ch, u) - B0;
                lvl = lvl > BW + 1 ? BW + 1 : lvl;
                bump(lvl, sgn);
            }
        }
    };
    // add (sgn > 0) or remove one bucket
    auto apply_bucket = [&](const TailBucket<NPERT>& b, int sgn) {
        const int cnt = cell_valid ? (int)(b.aux & 0x3FFu) : 0;
        // in the band <=> bin >= B0 <=> key >= (B0 + 1) << 7 <=> key > lim
        const unsigned lim = ((unsigned)(B0 + 1) << TAIL_POS_BITS) - 1u;
        const unsigned lim_rep = lim | (lim << 16);
        const tl_rsrc_t r = tl_make_rsrc(lists + (size_t)b.d * day_stride);
        int n_in = 0;
        // one list: count the keys of its first chunk that lie inside the band, bump them, go on to the next chunk while
        // a whole chunk was inside
        auto one_list = [&](const uint4& ch, int p) {
            const int n = cnt > 0 ? tl_count_above(ch, lim_rep) : 0;
            n_in += n;
            if (ctx_asm_bump) { if (sgn > 0) tl_bump_chunk<1>(ch, n, B0, BW + 1, col_lds); else tl_bump_chunk<-1>(ch, n, B0, BW + 1, col_lds); }
            else bump_chunk(ch, n, sgn, NPERT >= 3 ? 2 : 1);
            bool full = n == 8;
#pragma unroll
            for (int jj = 1; jj < 4; ++jj) {
                if (jj >= nch || __builtin_amdgcn_ballot_w64(full) == 0) break;
                const uint4 cj = tl_load_chunk(r, voff, (unsigned)(nch * p + jj) * chunk_row);
                const int nj = full ? tl_count_above(cj, lim_rep) : 0;
                n_in += nj;
                if (ctx_asm_bump) { if (sgn > 0) tl_bump_chunk<1>(cj, nj, B0, BW + 1, col_lds); else tl_bump_chunk<-1>(cj, nj, B0, BW + 1, col_lds); }
                else bump_chunk(cj, nj, sgn, 0);
                full = nj == 8;
            }
        };
#pragma unroll
        for (int p = 0; p < NPF; ++p)
            if (p < NPER) one_list(b.c0[p], p);
        // the lists that were not prefetched: two first chunks in flight at a time
#pragma nounroll
        for (int p = NPF; p < NPER; p += 2) {  // uniform trip count
            const uint4 e0 = tl_load_chunk(r, voff, (unsigned)(nch * p) * chunk_row);
            const uint4 e1 = p + 1 < NPER ? tl_load_chunk(r, voff, (unsigned)(nch * (p + 1)) * chunk_row) : make_uint4(0, 0, 0, 0);
            one_list(e0, p);
            if (p + 1 < NPER) one_list(e1, p + 1);
        }
        const int below = cnt - n_in;  // everything under the band
        if (below > 0) __hip_atomic_fetch_add((lds_u32*)(size_t)col_lds, (unsigned)(sgn > 0 ? below : -below), __ATOMIC_RELAXED,
                                              __HIP_MEMORY_SCOPE_WORKGROUP);
        mytot += sgn > 0 ? cnt : -cnt;
    };
    // in-place inclusive prefix sum over the lane's column, packed: (a0, a1) -> (a0, a0 + a1) + running total in both halves
    auto prefix = [&]() -> unsigned {
        unsigned run_rep = 0;
#pragma unroll
        for (int i = 0; i < (NLP + 1) / 2; ++i) {
            uint2 w = mycol2[i];
            w.x = pk_add_u16(w.x + (w.x << 16), run_rep);
            run_rep = __builtin_amdgcn_perm(w.x, w.x, 0x03020302u);
            w.y = pk_add_u16(w.y + (w.y << 16), run_rep);
            run_rep = __builtin_amdgcn_perm(w.y, w.y, 0x03020302u);
            mycol2[i] = w;
        }
        return run_rep & 0xFFFFu;
    };
    auto unprefix = [&]() {
        unsigned prev = 0;
#pragma unroll
        for (int i = 0; i < (NLP + 1) / 2; ++i) {
            const uint2 w = mycol2[i];
            uint2 o;
            o.x = pk_sub_u16(w.x, __builtin_amdgcn_alignbit(w.x, prev, 16));  // (c0 - prev_hi, c1 - c0)
            o.y = pk_sub_u16(w.y, __builtin_amdgcn_alignbit(w.y, w.x, 16));
            prev = w.y;
            mycol2[i] = o;
        }
    };
    auto wrap = [&](int d) { return ((d % NDOY) + NDOY) % NDOY; };
    // columns of day d from scratch for the current band
    auto rebuild = [&](int d) {
        for (int r = 0; r < TT_LS / 2; ++r) mycol2[r] = make_uint2(0u, 0u);
        mytot = 0;
        TailBucket<NPERT> cur = load_bucket(wrap(d - pd));
        for (int o = -pd + 1; o <= pd; ++o) {
            const TailBucket<NPERT> nxt = load_bucket(wrap(d + o));
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

    // ---- first band of the block: every cell's own bucket of the first day gives an estimate of its quantile bin (the
    // largest of the keys of rank ceil((1 - q) n / lists) in its lists); the band is centred on the tile's range of
    // estimates.  A bad estimate only costs passes below, never a wrong result.
    if (t == 0) {
        s_est_min = 0x7fffffff;
        s_est_max = -1;
        s_lo[0] = s_lo[1] = s_hi[0] = s_hi[1] = 0;
        s_iumin[0] = s_iumin[1] = 0x7fffffff;
        s_iumax[0] = s_iumax[1] = -1;
    }
    __syncthreads();
    {
        const TailBucket<NPERT> b = load_bucket(d_begin);
        const int cnt = cell_valid ? (int)(b.aux & 0x3FFu) : 0;
        if (cnt > 0) {
            int rank = (int)ceil((1.0 - q) * (double)cnt / (double)NPER);
            rank = rank < 1 ? 1 : (rank > 8 ? 8 : rank);
            unsigned key = 0;
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (u == rank - 1) key = tl_key(b.c0[0], u);
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
    TailBucket<NPERT> pin, pout;
    pin.d = pout.d = 0;
    pin.aux = pout.aux = 0;
#pragma unroll
    for (int p = 0; p < NPF; ++p) pin.c0[p] = pout.c0[p] = make_uint4(0, 0, 0, 0);
#ifdef MAREX_STAMPS
    unsigned long long st_p1 = 0, st_p2 = 0, st_bar = 0;
#define STAMP() __builtin_amdgcn_s_memtime()
#endif
    for (int dd = 0; dd < ndays; ++dd) {
        const int d = d_begin + dd;
        const int dpar = dd & 1;
#ifdef MAREX_STAMPS
        const unsigned long long tA = STAMP();
#endif
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
#ifdef MAREX_STAMPS
        const unsigned long long tB = STAMP();
        st_p1 += tB - tA;
#endif
        // ---------------- P2: quantile level of every output cell; stragglers move the band
        bool resolved = !(is_out && !land);
        int tried_lo = B0, tried_hi = B0;
        bool excursion = false;
        for (int pass = 0;; ++pass) {
            const int par = pass & 1;
#ifdef MAREX_STAMPS
            const unsigned long long tC0 = STAMP();
#endif
            __syncthreads();
#ifdef MAREX_STAMPS
            const unsigned long long tC = STAMP();
            st_bar += tC - tC0;
#endif
            if (t == 0) {
                s_lo[par ^ 1] = 0;
                s_hi[par ^ 1] = 0;
                if (pass == 0) {
                    s_iumin[dpar ^ 1] = 0x7fffffff;
                    s_iumax[dpar ^ 1] = -1;
                }
            }
            int iu_lo = 0x7fffffff, iu_hi = -1;  // this lane's contribution to the day's range of quantile bins
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
                        iu_lo = iu_hi = iu;
                        resolved = true;
                    }
                } else {
                    thr[(size_t)d * C + cell] = nan_f();  // empty window
                    resolved = true;
                }
            }
            if (__builtin_amdgcn_ballot_w64(iu_hi >= 0) != 0) {  // one pair of LDS atomics per wave, not one per lane on one address
                const int wlo = wave_min_i32(iu_lo), whi = wave_max_i32(iu_hi);
                if ((t & 63) == 0) {
                    atomicMin(&s_iumin[dpar], wlo);
                    atomicMax(&s_iumax[dpar], whi);
                }
            }
#ifdef MAREX_STAMPS
            const unsigned long long tD = STAMP();
            st_p2 += tD - tC;
#endif
            __syncthreads();
#ifdef MAREX_STAMPS
            st_bar += STAMP() - tD;
#endif
            const int lo = s_lo[par], hi = s_hi[par];
            if (!lo && !hi) break;
            if (pass >= pass_limit) {  // cannot happen while the bands of the passes overlap (static_assert above), but a wave must
                if (!resolved) {       // always reach the end of the walk: the output is counted as unresolved (the host raises) and NaN
                    thr[(size_t)d * C + cell] = nan_f();
                    atomicAdd(&stats->n_unresolved, 1u);
                }
                break;
            }
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
    }
    if ((t & 63) == 0) {
        if (kmin != 0xFFFFFFFFu) atomicMin(&stats->min_key, kmin);
        if (kmax != 0u) atomicMax(&stats->max_key, kmax);
        if (nlow) atomicAdd(&stats->n_too_low, nlow);
        if (nhigh) atomicAdd(&stats->n_too_high, nhigh);
    }
#ifdef MAREX_STAMPS
#ifdef MAREX_STAMPS_WAVE  // phase timers of wave MAREX_STAMPS_WAVE (-1: a different wave per tile) instead of wave 0
    const int st_wave = MAREX_STAMPS_WAVE >= 0 ? MAREX_STAMPS_WAVE : (int)((blockIdx.x + blockIdx.y) % (NT / 64));
    if (dbg && t == st_wave * 64) {
#else
    if (dbg && t == 0) {
#endif
        atomicAdd(&dbg[5], st_p1);
        atomicAdd(&dbg[6], st_p2);
        atomicAdd(&dbg[7], st_bar);
    }
#endif
    if (dbg && t == 0) {
        atomicAdd(&dbg[0], n_rebuild);
        atomicAdd(&dbg[2], n_pass);
        atomicAdd(&dbg[3], (unsigned long long)ndays);
    }
}

// ------------------------------------------------------------------------------------------------
// K_T without spatial pooling (window_spatial_hobday = 1: unstructured meshes, or a gridded field asked for it): every cell
// is on its own, so there is nothing to share between lanes -- no tiles, no halo, no barriers.  A lane owns one cell and
// walks a block of days.  The first chunk (the 8 largest keys) of every list of the wd buckets in its window waits in a
// per-wave LDS ring (read back only by the lane itself).
// The quantile bin of a day is found by bisection on the bin index:
//   iu = number of bins b with cs[b] <= qpos,  cs[b] = tot - #(samples with bin > b)           (detect.py:2510-2527)
// a probe counts the ring's keys at or above the probe's first key, 12 instructions per list.  A first chunk gives a LOWER
// bound of a list's count; that is enough whenever the bound already decides the comparison (a high quantile needs a handful of
// samples of many lists), and the rare probe it does not decide -- some list with its whole first chunk at or above the probe while
// the total is still short -- is recounted from the full lists.
// ------------------------------------------------------------------------------------------------
template <int NSC>  // NSC > 0: the ring has exactly NSC slots and a day's probes count from a REGISTER copy of it (one LDS read of
                    // the ring per day instead of one per probe: the probes of a day form a dependent chain, LDS latency included)
__global__ void __launch_bounds__(64)
k_thr_cells(const uint4* __restrict__ lists, const unsigned* __restrict__ aux, int NPER, int nch, const float* __restrict__ anom,
            long C, long c0, long c1, int nblk, const float* __restrict__ centres, int nb, double q, int wd, float lower_bound,
            float upper_bound, float* __restrict__ thr, marex_thr_stats* __restrict__ stats, unsigned long long* __restrict__ dbg) {
    extern __shared__ uint4 ring[];  // [wd * NPER rounded up to a multiple of 4][64]; the padding stays zero (no keys)
    const int lane = (int)threadIdx.x;
    const long cell_raw = c0 + (long)blockIdx.x * 64 + lane;
    const bool valid = cell_raw < c1;
    const long cell = valid ? cell_raw : c1 - 1;
    const int d_begin = (int)blockIdx.y * NDOY / nblk, d_end = ((int)blockIdx.y + 1) * NDOY / nblk;
    const bool land = !(anom[cell] == anom[cell]);  // first kept anomaly row (detect.py:2704)
    if (__builtin_amdgcn_ballot_w64(valid && !land) == 0) {
        if (valid)
            for (int d = d_begin; d < d_end; ++d) thr[(size_t)d * C + cell] = nan_f();
        return;
    }
    const int pd = wd / 2, nslot = (wd * NPER + 3) & ~3;
    uint4* mine = ring + lane;  // + slot * 64
    for (int k = wd * NPER; k < nslot; ++k) mine[k * 64] = make_uint4(0, 0, 0, 0);
    const unsigned voff = (unsigned)cell * 16u, chunk_row = (unsigned)C * 16u;
    const size_t day_stride = (size_t)NPER * nch * (size_t)C;
    auto wrap = [](int d) { return d < 0 ? d + NDOY : (d >= NDOY ? d - NDOY : d); };
    auto cnt_of = [&](int day) { return (int)(aux[(size_t)day * C + cell] & 0x3FFu); };
    auto load_day = [&](int day, int k) {  // bucket `day` into ring position k % wd: the first chunk of each of its lists
        const tl_rsrc_t r = tl_make_rsrc(lists + (size_t)day * day_stride);
        for (int p = 0; p < NPER; ++p) mine[((k % wd) * NPER + p) * 64] = tl_load_chunk(r, voff, (unsigned)(nch * p) * chunk_row);
    };
    int tot = 0;
    for (int o = 0; o < wd; ++o) {
        const int day = wrap(d_begin - pd + o);
        load_day(day, o);
        tot += cnt_of(day);
    }
    // lower bound of #(keys >= first key of bin b) from the ring
    uint4 cache[NSC > 0 ? NSC : 1];
    auto count_ring = [&](int b) {
        const unsigned lim = ((unsigned)(b + 1) << TAIL_POS_BITS) - 1u, lim_rep = lim | (lim << 16);
        unsigned acc0 = 0, acc1 = 0;
        if (NSC > 0) {
#pragma unroll
            for (int k = 0; k < NSC; ++k) {
                acc0 += pk_min1_u16(pk_sub_sat_u16(cache[k].x, lim_rep)) + pk_min1_u16(pk_sub_sat_u16(cache[k].z, lim_rep));
                acc1 += pk_min1_u16(pk_sub_sat_u16(cache[k].y, lim_rep)) + pk_min1_u16(pk_sub_sat_u16(cache[k].w, lim_rep));
            }
        } else
        for (int k = 0; k < nslot; k += 4) {  // four lists in flight per trip
            const uint4 c[4] = {mine[k * 64], mine[(k + 1) * 64], mine[(k + 2) * 64], mine[(k + 3) * 64]};
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                acc0 += pk_min1_u16(pk_sub_sat_u16(c[u].x, lim_rep)) + pk_min1_u16(pk_sub_sat_u16(c[u].z, lim_rep));
                acc1 += pk_min1_u16(pk_sub_sat_u16(c[u].y, lim_rep)) + pk_min1_u16(pk_sub_sat_u16(c[u].w, lim_rep));
            }
        }
        const unsigned acc = acc0 + acc1;  // halves stay far below 65536
        return (int)((acc & 0xFFFFu) + (acc >> 16));
    };
    // exact count from every chunk of every list of the window of day d (rare)
    auto count_exact = [&](int d, int b) {
        const unsigned lim = ((unsigned)(b + 1) << TAIL_POS_BITS) - 1u, lim_rep = lim | (lim << 16);
        int n = 0;
        for (int o = -pd; o <= pd; ++o) {
            const tl_rsrc_t r = tl_make_rsrc(lists + (size_t)wrap(d + o) * day_stride);
            for (int p = 0; p < NPER; ++p)
                for (int jj = 0; jj < nch; ++jj) n += tl_count_above(tl_load_chunk(r, voff, (unsigned)(nch * p + jj) * chunk_row), lim_rep);
        }
        return n;
    };
    unsigned kmin = 0xFFFFFFFFu, kmax = 0u, nlow = 0u, nhigh = 0u;
    unsigned long long n_exact = 0;
    for (int d = d_begin; d < d_end; ++d) {
        if (d > d_begin) {  // the window moves on: bucket d + pd takes the ring position of bucket d - pd - 1
            const int din = wrap(d + pd), dout = wrap(d - pd - 1);
            load_day(din, d - d_begin - 1);
            tot += cnt_of(din) - cnt_of(dout);
        }
        if (NSC > 0) {
#pragma unroll
            for (int k = 0; k < NSC; ++k) cache[k] = mine[k * 64];
        }
        // kfull = largest last key of a first chunk: a probe at or below it may be missing keys the ring does not hold;
        // khead = largest key of the window: no sample lies in a bin above its bin
        // ktail = smallest largest-key of a non-empty list: every such list has a key at or above it
        unsigned kfull = 0, khead = 0, ktail = 0xFFFFu;
        int nlists = 0;
        for (int k = 0; k < nslot; k += 4) {
            const uint4 c[4] = {mine[k * 64], mine[(k + 1) * 64], mine[(k + 2) * 64], mine[(k + 3) * 64]};
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                kfull = c[u].w > kfull ? c[u].w : kfull;                                   // compares the high halves first
                const unsigned head = c[u].x & 0xFFFFu;
                khead = head > khead ? head : khead;
                ktail = (head != 0u && head < ktail) ? head : ktail;
                nlists += head != 0u ? 1 : 0;
            }
        }
        kfull >>= 16;
        float t32 = nan_f();
        if (tot > 0) {
            const double qpos = q * (double)tot;
            // f(b) = cs[b] <= qpos is true below iu and false from iu on; answer in (lo, hi].  hi starts at the bin of the
            // largest key (nothing above it: f is false there unless qpos >= tot, which the clip to nb - 1 covers).
            // n_lo = #(bin > lo) (a lower bound unless lo_exact), n_hi = #(bin > hi) (exact)
            int lo = -1, hi = nb - 1, n_lo = tot, n_hi = 0;
            if (khead != 0u && (double)tot > qpos) {
                const int bh = (int)(khead >> TAIL_POS_BITS) - 1;
                hi = bh < hi ? bh : hi;
            }
            bool lo_exact = true;
            // at least nlists samples lie in bins >= bin(ktail): if that already reaches tot - qpos, f is true below that bin
            if (nlists > 0 && (double)(tot - nlists) <= qpos) {
                const int bt = (int)(ktail >> TAIL_POS_BITS) - 2;  // f(bt) looks at bins > bt = bins >= bin(ktail)
                if (bt > lo && bt < hi) {
                    lo = bt;
                    n_lo = nlists;
                    lo_exact = false;
                }
            }
            while (__builtin_amdgcn_ballot_w64(hi - lo > 1) != 0) {
                const bool go = hi - lo > 1;
                const int mid = go ? (lo + hi) >> 1 : hi;
                int n = count_ring(mid + 1);
                const unsigned first = (unsigned)(mid + 2) << TAIL_POS_BITS;  // first key of bin mid + 1
                bool exact = !(first <= kfull);
                bool f = (double)(tot - n) <= qpos;
                if (__builtin_amdgcn_ballot_w64(go && !f && !exact) != 0) {  // the bound does not decide: count everything
                    const int ne = count_exact(d, mid + 1);
                    if (go && !f && !exact) {
                        n = ne;
                        exact = true;
                        f = (double)(tot - n) <= qpos;
                        ++n_exact;
                    }
                }
                if (go) {
                    if (f) {
                        lo = mid;
                        n_lo = n;
                        lo_exact = exact;
                    } else {
                        hi = mid;
                        n_hi = n;
                    }
                }
            }
            const int iu = hi;
            if (__builtin_amdgcn_ballot_w64(iu > 0 && !lo_exact) != 0) {
                const int ne = count_exact(d, iu);
                if (iu > 0 && !lo_exact) {
                    n_lo = ne;
                    ++n_exact;
                }
            }
            const int il = iu > 0 ? iu - 1 : 0;
            const int cs_iu = tot - n_hi, cs_il = iu > 0 ? tot - n_lo : cs_iu;
            const int diff = cs_iu - cs_il;
            const double frac = diff > 0 ? (qpos - (double)cs_il) / (double)diff : 0.5;
            const float dc = centres[iu] - centres[il];
            const double prod = frac * (double)dc;
            t32 = (float)((double)centres[il] + prod);
            if (iu == 0) t32 = centres[0];
            if (valid && !land) {
                const unsigned key = ordered_key(t32);
                kmin = key < kmin ? key : kmin;
                kmax = key > kmax ? key : kmax;
                if (t32 > upper_bound) ++nhigh;
                if (t32 < lower_bound) ++nlow;
            }
            if (t32 < lower_bound) t32 = lower_bound;
        }
        if (valid) thr[(size_t)d * C + cell] = land ? nan_f() : t32;
    }
    for (int sft = 32; sft > 0; sft >>= 1) {
        const unsigned a = __shfl_down(kmin, sft, 64), b = __shfl_down(kmax, sft, 64);
        kmin = a < kmin ? a : kmin;
        kmax = b > kmax ? b : kmax;
        nlow += __shfl_down(nlow, sft, 64);
        nhigh += __shfl_down(nhigh, sft, 64);
        n_exact += __shfl_down(n_exact, sft, 64);
    }
    if (lane == 0) {
        if (kmin != 0xFFFFFFFFu) atomicMin(&stats->min_key, kmin);
        if (kmax != 0u) atomicMax(&stats->max_key, kmax);
        if (nlow) atomicAdd(&stats->n_too_low, nlow);
        if (nhigh) atomicAdd(&stats->n_too_high, nhigh);
        if (dbg && n_exact) atomicAdd(&dbg[1], n_exact);
    }
}

extern "C" int marex_hobday_thresholds_tails_f32(marex_ctx* ctx, const void* lists, const uint32_t* aux, int list_rows,
                                                 const float* anom, int64_t T_out, int64_t C, int ny, int nx, int max_bucket,
                                                 const float* centres,
                                                 int nb, double q, int wd, int ws, float lower_bound, float upper_bound, int row0,
                                                 int row1, float* thr_doy_major, marex_thr_stats* stats) {
    if (!ctx) return -1;
    if (!lists || !aux || !anom || !centres || !thr_doy_major || !stats || T_out <= 0 || C <= 0)
        return fail(ctx, -1, "marex_hobday_thresholds_tails_f32: null pointer or empty shape");
    if (wd < 3 || wd > 365 || (wd & 1) == 0)
        return fail(ctx, -1, "marex_hobday_thresholds_tails_f32: window_days_hobday must be odd and in 3..365");
    if (ws < 1 || (ws & 1) == 0) return fail(ctx, -1, "marex_hobday_thresholds_tails_f32: window_spatial_hobday must be odd");
    if (!(q > 0.0 && q <= 1.0)) return fail(ctx, -1, "marex_hobday_thresholds_tails_f32: q must be in (0, 1]");
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
        (int64_t)max_bucket * wd * ws * ws > 65535 || C > (1 << 24))
        return fail(ctx, -4, "marex_hobday_thresholds_tails_f32: shape outside the tail kernel (nb <= %d, buckets <= %d rows, "
                             "ws <= 7, pooled window <= 65535 samples)", TAIL_MAX_NB, TAIL_MAX_BUCKET);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    int NPER, nch;
    if (!tails_geometry(max_bucket, list_rows, NPER, nch))
        return fail(ctx, -4, "marex_hobday_thresholds_tails_f32: %d rows per bucket in lists of %d", max_bucket, list_rows);
    // no spatial pooling: one lane per cell, no tiles -- while the window is a dozen lists (measured on 0.5 M cells: 12 slots 7.9 ms
    // against 24.9 ms for the tile kernel; 24 slots 17.7 against 13.3: a probe's cost grows with the lists, the tile kernel's
    // column update does not).  THR_CELLS=2 forces it for any window that fits the LDS.
    const int cells_opt = ctx_opt(ctx, "THR_CELLS", 1);
    if (p == 0 && cells_opt && (cells_opt == 2 ? wd * NPER <= 128 : wd * NPER <= 16)) {
        const long c0 = ny > 0 ? (long)row0 * nx : 0, c1 = ny > 0 ? (long)row1 * nx : (long)C;
        const long ncg = (c1 - c0 + 63) / 64;
        int nblk = (int)((32L * 4 * device_cus(ctx) + ncg - 1) / ncg);  // several rounds of waves (an even finish); every block re-reads wd - 1 buckets
        nblk = nblk < 1 ? 1 : (nblk > 12 ? 12 : nblk);
        nblk = ctx_opt(ctx, "THR_CELLS_BLOCKS", nblk);
        LaunchTimer lt(ctx, MAREX_K_THRESHOLDS);
        const size_t lds = (size_t)((wd * NPER + 3) & ~3) * 64 * sizeof(uint4);
        const int nslot = (wd * NPER + 3) & ~3;
        const bool cached = nslot == 12 && ctx_opt(ctx, "THR_CELLS_CACHE", 1);  // the default 11-day window of single-list buckets
        if (lds > 64 * 1024) HIP_TRY(ctx, hipFuncSetAttribute((const void*)k_thr_cells<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(cached ? k_thr_cells<12> : k_thr_cells<0>, dim3((unsigned)ncg, (unsigned)nblk), dim3(64), lds, ctx->stream,
                           reinterpret_cast<const uint4*>(lists), aux, NPER, nch, anom, (long)C, c0, c1, nblk, centres, nb, q, wd,
                           lower_bound, upper_bound, thr_doy_major, stats, ctx_debug_counters(ctx));
        HIP_TRY(ctx, hipGetLastError());
        return 0;
    }
    if (NPER > 6)
        return fail(ctx, -4, "marex_hobday_thresholds_tails_f32: %d rows per bucket in lists of %d: more than 6 lists", max_bucket, list_rows);
    const int tile_pref = ctx_opt(ctx, "THR_TILE", (ny > 0 && p > 0 && max_bucket >= 24) ? 32 : 16);
    // 3216 (experiments): 32 wide x 16 tall tiles on 512 threads, P = 2 -- half a CU's LDS and registers per workgroup, so that
    // workgroups of ANOTHER kernel (the anomaly kernel of the neighbouring stream) can share the CU; THR_LDS_PAD (KiB of unused
    // dynamic LDS per workgroup) keeps a second tile of this kernel off the CU
    const bool half = (ny > 0 && p == 2) && tile_pref == 3216 && (row1 - row0) >= 12 && nx >= 16;
    const bool big = !half && (ny > 0 && p > 0) && tile_pref == 32 && (row1 - row0) >= 16 && nx >= 16;
    const int NT = big ? 1024 : (half ? 512 : 256);
    int TR = (ny > 0 && p > 0) ? (big ? 32 : 16) : 1, TC = NT / TR;
    const size_t lds_pad = (size_t)ctx_opt(ctx, "THR_LDS_PAD", 0) * 1024;
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
        const long slots = cus * (big ? 1 : (half ? 2 : 4));
        long best = -1;
        for (int d = 16; d <= NDOY; ++d) {
            const long blocks = tiles * ((NDOY + d - 1) / d);
            const long cost = ((blocks + slots - 1) / slots) * (d + wd);
            if (best < 0 || cost < best) best = cost, Dd = d;
        }
    }
    dim3 grid((unsigned)(tiles_x * tiles_y), (unsigned)((NDOY + Dd - 1) / Dd));
    unsigned long long* dbg = ctx_debug_counters(ctx);
    const uint4* tl = reinterpret_cast<const uint4*>(lists);
    // straggler passes per day before the kernel gives up on an output (never reached: the tried bands overlap, so at most
    // nb / TT_STEP + 1 passes in either direction find any bin; option THR_PASS_LIMIT exists for the test of the error path)
    const int pass_limit = ctx_opt(ctx, "THR_PASS_LIMIT", 2 * (nb / TT_STEP) + 9);
#define MAREX_TT_ARGS tl, aux, NPER, nch, anom, (long)C, ny, nx, row0, row1, tiles_x, Dd, centres, nb, q, wd, lower_bound, upper_bound, thr_doy_major, stats, dbg, pass_limit
#define MAREX_TT_LAUNCH(PP, TCC, NTT, ...)                                                                                           \
    do {                                                                                                                             \
        if (NPER <= 1)                                                                                                               \
            hipLaunchKernelGGL((k_thr_tails<PP, TCC, NTT, 1, ##__VA_ARGS__>), grid, dim3(NTT), lds_pad, ctx->stream, MAREX_TT_ARGS); \
        else if (NPER <= 2)                                                                                                          \
            hipLaunchKernelGGL((k_thr_tails<PP, TCC, NTT, 2, ##__VA_ARGS__>), grid, dim3(NTT), lds_pad, ctx->stream, MAREX_TT_ARGS); \
        else /* two lists prefetched across the barrier, the others loaded at use */                                                 \
            hipLaunchKernelGGL((k_thr_tails<PP, TCC, NTT, 3, ##__VA_ARGS__>), grid, dim3(NTT), lds_pad, ctx->stream, MAREX_TT_ARGS); \
    } while (0)
    {
        LaunchTimer lt(ctx, MAREX_K_THRESHOLDS);
        if (TR == 1)
            MAREX_TT_LAUNCH(0, 256, 256);
        else if (half)
            MAREX_TT_LAUNCH(2, 32, 512, 16);
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
// Every sample at or above the threshold sits at the top of one of its bucket's lists: the keys at or above the
// threshold's bin are counted with packed arithmetic (they are a prefix of the sorted chunk); those above it are extremes,
// those in it are compared as numbers (a handful per wave and day).  Buckets holding values beyond the edge table (aux bit 15) are decided on
// the anomalies themselves (float4 rows, like the plain mask kernel).
// Lane = 4 consecutive cells: 16-byte key chunks of 4 cells are one contiguous 64-byte run, mask stores are 4 bytes.
// ------------------------------------------------------------------------------------------------
#ifndef MASK_WAVES
#define MASK_WAVES 5  // waves per SIMD the mask kernel is compiled for (96 VGPRs, a few spilled: 2.24 -> 2.03 ms; 6 is worse)
#endif
template <int NPERT>  // lists handled per group (their first chunks are in flight together)
__global__ void __launch_bounds__(256, MASK_WAVES)
k_mask_tails(const uint4* __restrict__ lists, const unsigned* __restrict__ aux, int NPER_all, int nch, const float* __restrict__ anom,
             const float* __restrict__ edges, int nb, const float* __restrict__ thr, const int* __restrict__ doy_start,
             const int* __restrict__ doy_rows, const long long* __restrict__ row_off, const long long* __restrict__ row_off_anom,
             long C, long c0, long c1, unsigned char* __restrict__ out, unsigned long long* __restrict__ n_true,
             unsigned long long* __restrict__ dbg) {
    const int nchunk = (int)gridDim.y;
    const int dA = (int)blockIdx.y * NDOY / nchunk, dB = ((int)blockIdx.y + 1) * NDOY / nchunk;
    // lanes are laid over the cells from the multiple of 256 below c0: a wave's row segment is then 256 bytes at a 256-byte
    // offset of the row (with two overlap rows of 1440 cells c0 = 2880 = 11.25 x 256, and every wave store straddled three
    // 128-byte lines: the 1.21 x write amplification of the 100-yr bands, round 3)
    const long c = (c0 & ~255L) + ((long)blockIdx.x * 256 + threadIdx.x) * 4;
    unsigned cnt_true = 0;
    unsigned long long n_slow = 0;
    if (c >= c0 && c < c1) {
        const float inv_width = (float)(nb - 1) / (edges[nb] - edges[1]);
        for (int d = dA; d < dB; ++d) {
            const int r0 = doy_start[d], nd = doy_start[d + 1] - r0;
            if (nd == 0) continue;
            const float4 th4 = *reinterpret_cast<const float4*>(thr + (size_t)d * C + c);
            const float tv[4] = {th4.x, th4.y, th4.z, th4.w};
            const uint4 ax = *reinterpret_cast<const uint4*>(aux + (size_t)d * C + c);
            const unsigned av[4] = {ax.x, ax.y, ax.z, ax.w};
            unsigned beyond = 0;  // some cell of the lane has samples beyond the table and a threshold that is a number
            unsigned bits[4][4];
            unsigned lim_ge[4];  // keys > lim_ge: bin >= the threshold's bin
            int kt[4];
            bool slow = false, walk[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                bits[i][0] = bits[i][1] = bits[i][2] = bits[i][3] = 0u;
                const bool isnum = tv[i] == tv[i];
                kt[i] = isnum ? digitize_bin(tv[i], edges, nb, inv_width) : nb;  // NaN threshold: nothing is extreme
                // values beyond the table are extremes of every threshold INSIDE the table (any threshold the histogram quantile
                // produces); more than two of them, or a threshold beyond the table itself: look at the values
                slow = slow || (isnum && (av[i] & TAIL_AUX_BEYOND) && ((av[i] & TAIL_AUX_MANY) || kt[i] >= nb));
                beyond |= (isnum && (av[i] & TAIL_AUX_BEYOND)) ? 1u : 0u;
                walk[i] = isnum && (av[i] & 0x3FFu) > 0 && kt[i] < nb;
                const unsigned lim = ((unsigned)(kt[i] + 1) << TAIL_POS_BITS) - 1u;
                lim_ge[i] = lim | (lim << 16);
            }
            // Keys of a chunk at or above the threshold's bin are a prefix of n_ge keys; the first n_gt of them lie in higher
            // bins (extremes), the rest sit IN the threshold's bin and are compared as numbers.  Those compares need a value
            // from HBM: the first such key of every (list, cell) is fetched for all of them at once (one round trip per
            // dayofyear instead of one per key); a second key in the same bin -- or a first chunk entirely above the
            // threshold -- is rare and sends the 4-cell group to the plain compare on the anomalies.
            auto set_prefix = [&](const uint4& q, int n, int i) {  // rows of the first n keys of q -> bits of cell i
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const bool on = n > u;
                    if (u >= 1 && __builtin_amdgcn_ballot_w64(on) == 0) break;
                    if (on) {
                        const int pos = (int)(tl_key(q, u) & (TAIL_MAX_BUCKET - 1));
#pragma unroll
                        for (int wi = 0; wi < 4; ++wi)
                            if ((pos >> 5) == wi) bits[i][wi] |= 1u << (pos & 31);
                    }
                }
            };
            if (__builtin_amdgcn_ballot_w64(!slow) != 0)
            for (int pg = 0; pg < NPER_all; pg += NPERT) {  // lists in groups of NPERT
                const int NPER = (NPER_all - pg) < NPERT ? (NPER_all - pg) : NPERT;
                // every list's first chunk of the 4 cells: 64 contiguous bytes per list, all loads in flight together
                uint4 ch[NPERT][4];
                const uint4* row0 = lists + (((size_t)d * NPER_all + pg) * nch) * C + c;
#pragma unroll
                for (int p = 0; p < NPERT; ++p)
#pragma unroll
                    for (int i = 0; i < 4; ++i) ch[p][i] = p < NPER ? row0[(size_t)(p * nch) * C + i] : make_uint4(0, 0, 0, 0);
            float cand_val[NPERT][4];
            int cand_pos[NPERT][4];
            unsigned redo = 0;  // (list, cell) pairs the batched pass could not finish
            {
#pragma unroll
                for (int p = 0; p < NPERT; ++p) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        cand_pos[p][i] = -1;
                        cand_val[p][i] = 0.f;
                        if (p < NPER && walk[i] && !slow) {
                            const unsigned lgt = lim_ge[i] + (0x10001u << TAIL_POS_BITS);  // keys > lgt: bin > the threshold's bin
                            const int n_ge = tl_count_above(ch[p][i], lim_ge[i]);
                            const int n_gt = tl_count_above(ch[p][i], lgt);
                            set_prefix(ch[p][i], n_gt, i);
                            if (n_ge == 8 || n_ge - n_gt > 1) redo |= 1u << (p * 4 + i);  // the list goes on / keys share the threshold's bin
                            if (n_ge - n_gt == 1) {  // key number n_gt: shift the sorted chunk down by n_gt keys
                                const unsigned w[4] = {ch[p][i].x, ch[p][i].y, ch[p][i].z, ch[p][i].w};
                                unsigned lo = w[0], hi = w[1];
#pragma unroll
                                for (int k = 1; k < 4; ++k)
                                    if ((n_gt >> 1) == k) lo = w[k], hi = k < 3 ? w[k + 1] : 0u;
                                const unsigned key = (n_gt & 1) ? (lo >> 16) : (lo & 0xFFFFu);
                                (void)hi;
                                cand_pos[p][i] = (int)(key & (TAIL_MAX_BUCKET - 1));
                            }
                        }
                    }
                }
                // all candidate values in flight together
#pragma unroll
                for (int p = 0; p < NPERT; ++p)
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        if (cand_pos[p][i] >= 0)
                            cand_val[p][i] = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(anom) + row_off_anom[r0 + cand_pos[p][i]] + (size_t)(c + i) * 4);
#pragma unroll
                for (int p = 0; p < NPERT; ++p)
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        if (cand_pos[p][i] >= 0 && cand_val[p][i] >= tv[i]) {
                            const int pos = cand_pos[p][i];
#pragma unroll
                            for (int wi = 0; wi < 4; ++wi)
                                if ((pos >> 5) == wi) bits[i][wi] |= 1u << (pos & 31);
                        }
                // the rare leftovers: walk the whole list key by key, one value fetch per key of the threshold's bin
                // (setting a bit twice is harmless)
                if (__builtin_amdgcn_ballot_w64(redo != 0u) != 0) {
#pragma unroll
                    for (int p = 0; p < NPERT; ++p)
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            if (__builtin_amdgcn_ballot_w64((redo >> (p * 4 + i)) & 1u) == 0) continue;
                            bool on = (redo >> (p * 4 + i)) & 1u;
                            for (int jj = 0; jj < nch; ++jj) {
                                if (__builtin_amdgcn_ballot_w64(on) == 0) break;
                                const uint4 q = on ? row0[(size_t)(p * nch + jj) * C + i] : make_uint4(0, 0, 0, 0);
#pragma unroll
                                for (int u = 0; u < 8; ++u) {
                                    const unsigned key = tl_key(q, u);
                                    const int bin = (int)(key >> TAIL_POS_BITS) - 1, pos = (int)(key & (TAIL_MAX_BUCKET - 1));
                                    on = on && bin >= kt[i];
                                    if (on) {
                                        bool ext = bin > kt[i];
                                        if (!ext) ext = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(anom) + row_off_anom[r0 + pos] + (size_t)(c + i) * 4) >= tv[i];
                                        if (ext) {
#pragma unroll
                                            for (int wi = 0; wi < 4; ++wi)
                                                if ((pos >> 5) == wi) bits[i][wi] |= 1u << (pos & 31);
                                        }
                                    }
                                }
                            }
                        }
                }
            }
            }  // list groups
            // samples beyond the table have no key but are extremes of every finite threshold: their positions come with aux
            if (__builtin_amdgcn_ballot_w64(beyond != 0u && !slow) != 0) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    if (!slow && tv[i] == tv[i] && (av[i] & TAIL_AUX_BEYOND)) {
                        const int p1 = (int)((av[i] >> 16) & 0x7Fu), p2 = (int)((av[i] >> 23) & 0x7Fu);
                        const bool two = (av[i] & TAIL_AUX_SECOND) != 0u;
#pragma unroll
                        for (int wi = 0; wi < 4; ++wi) {
                            if ((p1 >> 5) == wi) bits[i][wi] |= 1u << (p1 & 31);
                            if (two && (p2 >> 5) == wi) bits[i][wi] |= 1u << (p2 & 31);
                        }
                    }
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
                // rows in groups of 8: eight row offsets per scalar load batch, bit positions known at compile time
                unsigned char* obase = out + c;
#pragma unroll
                for (int wi = 0; wi < 4; ++wi) {
                    const unsigned b0 = bits[0][wi], b1 = bits[1][wi], b2 = bits[2][wi], b3 = bits[3][wi];
                    cnt_true += __popc(b0) + __popc(b1) + __popc(b2) + __popc(b3);
                    if (wi * 32 >= nd) break;
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const int rlo = wi * 32 + g * 8;
                        if (rlo >= nd) break;
                        const unsigned q0 = b0 >> (g * 8), q1 = b1 >> (g * 8), q2 = b2 >> (g * 8), q3 = b3 >> (g * 8);
                        const long long* ro = row_off + r0 + rlo;
                        if (rlo + 8 <= nd) {
#pragma unroll
                            for (int rr = 0; rr < 8; ++rr) {
                                const unsigned m = ((q0 >> rr) & 1u) | (((q1 >> rr) & 1u) << 8) | (((q2 >> rr) & 1u) << 16) | (((q3 >> rr) & 1u) << 24);
                                __builtin_nontemporal_store(m, reinterpret_cast<unsigned*>(obase + ro[rr]));
                            }
                        } else {
                            for (int rr = 0; rr < nd - rlo; ++rr) {
                                const unsigned m = ((q0 >> rr) & 1u) | (((q1 >> rr) & 1u) << 8) | (((q2 >> rr) & 1u) << 16) | (((q3 >> rr) & 1u) << 24);
                                __builtin_nontemporal_store(m, reinterpret_cast<unsigned*>(obase + ro[rr]));
                            }
                        }
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

extern "C" int marex_mask_ge_doy_tails_f32(marex_ctx* ctx, const void* lists, const uint32_t* aux, int list_rows, int max_bucket,
                                           const float* anom,
                                           const float* edges, int nb, const float* thr_doy_major, const int32_t* doy_start,
                                           const int32_t* doy_rows, int64_t T_out, int64_t C, int64_t c0, int64_t c1,
                                           uint8_t* extreme, unsigned long long* n_true) {
    if (!ctx) return -1;
    if (!lists || !aux || !anom || !edges || !thr_doy_major || !doy_start || !doy_rows || !extreme || T_out <= 0 || C <= 0 || nb < 4)
        return fail(ctx, -1, "marex_mask_ge_doy_tails_f32: null pointer or empty shape");
    if (c0 < 0 || c1 > C || c0 >= c1) return fail(ctx, -1, "marex_mask_ge_doy_tails_f32: need 0 <= c0 < c1 <= C");
    int NPER, nch;
    if (!tails_geometry(max_bucket, list_rows, NPER, nch))
        return fail(ctx, -4, "marex_mask_ge_doy_tails_f32: buckets must hold 1..%d rows, lists 8..32 rows", TAIL_MAX_BUCKET);
    const bool vec = (C % 4 == 0) && (c0 % 4 == 0) && (c1 % 4 == 0) && (((uintptr_t)anom | (uintptr_t)thr_doy_major) % 16 == 0) &&
                     ((uintptr_t)extreme % 4 == 0) && ((uintptr_t)aux % 16 == 0) && nb < 0x7fff;
    if (!vec)  // shapes the 4-cell kernel does not cover: the plain compare on the anomalies
        return marex_mask_ge_doy_f32(ctx, anom, thr_doy_major, doy_start, doy_rows, T_out, C, c0, c1, extreme, n_true);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    {
        LaunchTimer lt(ctx, MAREX_K_MASK);
        const unsigned ncb = (unsigned)(((c1 - (c0 & ~255L)) / 4 + 255) / 256);
        // about 16 000 workgroups (measured on 100-yr bands of 90 / 120 rows: 33 / 25 chunks 2.25 / 2.90 ms, 61 chunks 2.12 / 2.79 ms,
        // 122 chunks 2.11 ms; the whole pass 125.7 -> 124.3 ms)
        unsigned chunks = (16384 + ncb - 1) / ncb;
        chunks = chunks < MASK_DOY_CHUNKS ? MASK_DOY_CHUNKS : (chunks > 61 ? 61 : chunks);
        {
            const int forced = ctx_opt(ctx, "MASK_CHUNKS", 0);  // dayofyear chunks of the grid (experiments)
            if (forced >= 1 && forced <= 366) chunks = (unsigned)forced;
        }
        unsigned long long* dbg = ctx_debug_counters(ctx);
        const uint4* tl = reinterpret_cast<const uint4*>(lists);
        // byte offset of every kept row of the MASK array (one byte per cell) in dayofyear order
        const size_t need = (size_t)T_out * sizeof(long long);
        if (need > ctx->row_off_mask_bytes) {
            if (ctx->row_off_mask) HIP_TRY(ctx, hipFree(ctx->row_off_mask));
            ctx->row_off_mask = nullptr;
            ctx->row_off_mask_bytes = 0;
            HIP_TRY(ctx, hipMalloc((void**)&ctx->row_off_mask, need));
            ctx->row_off_mask_bytes = need;
        }
        hipLaunchKernelGGL(k_row_offsets, dim3((unsigned)((T_out + 255) / 256)), dim3(256), 0, ctx->stream, doy_rows, (long)T_out, (long)C,
                           ctx->row_off_mask, 1);
        const size_t need4 = (size_t)T_out * sizeof(long long);
        if (need4 > ctx->row_off_bytes) {
            if (ctx->row_off) HIP_TRY(ctx, hipFree(ctx->row_off));
            ctx->row_off = nullptr;
            ctx->row_off_bytes = 0;
            HIP_TRY(ctx, hipMalloc((void**)&ctx->row_off, need4));
            ctx->row_off_bytes = need4;
        }
        hipLaunchKernelGGL(k_row_offsets, dim3((unsigned)((T_out + 255) / 256)), dim3(256), 0, ctx->stream, doy_rows, (long)T_out, (long)C,
                           ctx->row_off, (int)sizeof(float));
#define MAREX_MT_ARGS tl, aux, NPER, nch, anom, edges, nb, thr_doy_major, doy_start, doy_rows, ctx->row_off_mask, ctx->row_off, (long)C, (long)c0, (long)c1, extreme, n_true, dbg
        const int grp = ctx_opt(ctx, "MASK_GROUP", 0);  // lists handled at once (experiments: 1, 2, 3)
        if (NPER <= 1 || grp == 1)
            hipLaunchKernelGGL(k_mask_tails<1>, dim3(ncb, chunks), dim3(256), 0, ctx->stream, MAREX_MT_ARGS);
        else if ((NPER != 3 && grp != 3) || grp == 2)  // pairs of lists: 117 VGPRs = four waves per SIMD (three lists: 136)
            hipLaunchKernelGGL(k_mask_tails<2>, dim3(ncb, chunks), dim3(256), 0, ctx->stream, MAREX_MT_ARGS);
        else  // exactly 3 lists: one group
            hipLaunchKernelGGL(k_mask_tails<3>, dim3(ncb, chunks), dim3(256), 0, ctx->stream, MAREX_MT_ARGS);
#undef MAREX_MT_ARGS
    }
    HIP_TRY(ctx, hipGetLastError());
    return 0;
}
