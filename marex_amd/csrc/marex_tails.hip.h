#pragma once
// marex_tails.hip.h -- "tails": the upper end of every (cell, dayofyear) histogram as a short sorted list.
//
// The reference counts every anomaly into a dense (dayofyear, bin) histogram per cell and only ever asks it for one
// high quantile (detect.py:2638-2648, 2465-2559).  For q >= 0.6 the answer is decided by the few largest samples of the
// dayofyear buckets in the pooled window; everything below only has to be COUNTED.  A tail is that upper end:
//
//   key   = ((bin + 1) << 7) | pos      16 bits; bin = np.digitize(anom, edges) - 1 (< nb <= 511), pos = position of the
//                                       sample inside its dayofyear bucket (rows doy_start[d] + pos, < 128); 0 = empty
//   tails[d][j][c] (uint4 = 8 keys)     chunk j of the K keys of bucket (dayofyear d + 1, cell c), the K LARGEST keys of
//                                       the bucket sorted descending (chunk-major: one 16-byte load per lane and chunk,
//                                       contiguous across consecutive cells)
//   aux[d][c] (uint16)                  bits 0..9: samples of the bucket with a valid bin (bin < nb, the ones the
//                                       reference's histogram counts); bit 15: the bucket holds a non-NaN value at or
//                                       beyond the last edge (only the mask has to look at it)
//
// Keys are unique inside a bucket (pos), so "the samples that are not in the tail" are exactly those with a key below
// the K-th one: a consumer that needs them (tail exhausted while still inside its band of levels) re-reads the
// bucket's anomalies and takes the keys below the last tail key -- no flags, no second data structure.
//
// The sorting networks below work on PACKED PAIRS: one 32-bit register holds the keys of two independent buckets
// (two neighbouring dayofyears of one cell), v_pk_max_u16 / v_pk_min_u16 order both at once.
#include <stdint.h>

#if defined(__HIPCC__) || defined(__CUDACC__)
#define MAREX_HD __host__ __device__ __forceinline__
#else
#define MAREX_HD inline
#endif

#define TAIL_POS_BITS 7
#define TAIL_MAX_BUCKET 128  // pos < 2^7
#define TAIL_MAX_NB 511      // bin + 1 < 2^9

typedef unsigned short marex_us2 __attribute__((ext_vector_type(2)));

MAREX_HD unsigned pk_max_u16(unsigned a, unsigned b) {
    return __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(marex_us2, a), __builtin_bit_cast(marex_us2, b)));
}
MAREX_HD unsigned pk_min_u16(unsigned a, unsigned b) {
    return __builtin_bit_cast(unsigned, __builtin_elementwise_min(__builtin_bit_cast(marex_us2, a), __builtin_bit_cast(marex_us2, b)));
}
// compare-exchange, descending: afterwards a >= b in both halves
MAREX_HD void pk_cx(unsigned& a, unsigned& b) {
    const unsigned hi = pk_max_u16(a, b), lo = pk_min_u16(a, b);
    a = hi;
    b = lo;
}

// v[0..N) bitonic (first descending then ascending, or any rotation-free bitonic sequence) -> sorted descending
template <int N>
MAREX_HD void bitonic_merge_desc(unsigned (&v)[N]) {
#pragma unroll
    for (int half = N / 2; half >= 1; half >>= 1) {
#pragma unroll
        for (int i = 0; i < N; ++i)
            if ((i & half) == 0) pk_cx(v[i], v[i | half]);
    }
}

// any order -> sorted descending (bitonic sorting network, N a power of two)
template <int N>
MAREX_HD void bitonic_sort_desc(unsigned (&v)[N]) {
#pragma unroll
    for (int size = 2; size <= N; size <<= 1) {
        // first step of every merge compares i with its mirror inside the block of `size` (makes the halves bitonic)
#pragma unroll
        for (int i = 0; i < N; ++i) {
            const int j = i ^ (size - 1);
            if (j > i) pk_cx(v[i], v[j]);
        }
#pragma unroll
        for (int half = size / 4; half >= 1; half >>= 1) {
#pragma unroll
            for (int i = 0; i < N; ++i)
                if ((i & half) == 0) pk_cx(v[i], v[i | half]);
        }
    }
}

// top[0..K) sorted descending, nw[0..B) sorted descending (B <= K, both powers of two or K a multiple of B):
// top <- the K largest of the union, sorted descending.  nw is destroyed.
template <int K, int B>
MAREX_HD void tail_merge(unsigned (&top)[K], unsigned (&nw)[B]) {
    static_assert(B <= K, "batch larger than the tail");
    // the first K - B entries of top stay in the result whatever nw holds (at most B new keys can pass them); the last
    // B compete with nw: a half-cleaner of two sorted runs leaves the B largest, as a bitonic sequence
    unsigned cand[B];
#pragma unroll
    for (int i = 0; i < B; ++i) cand[i] = pk_max_u16(top[K - B + i], nw[B - 1 - i]);
    bitonic_merge_desc<B>(cand);
    if (K == B) {
#pragma unroll
        for (int i = 0; i < B; ++i) top[i] = cand[i];
        return;
    }
    // K > B: merge the safe prefix (sorted) with cand (sorted): prefix ++ reverse(cand) is bitonic.  For K = 2B this is
    // a plain bitonic merge; for larger K only the last 2B entries can change order beyond position K - 2B ... keep it
    // simple and exact: merge everything (K is 16 or 32 here).
    unsigned all[K];
#pragma unroll
    for (int i = 0; i < K - B; ++i) all[i] = top[i];
#pragma unroll
    for (int i = 0; i < B; ++i) all[K - B + i] = cand[B - 1 - i];
    bitonic_merge_desc<K>(all);
#pragma unroll
    for (int i = 0; i < K; ++i) top[i] = all[i];
}

MAREX_HD unsigned tail_key(int bin, int pos) { return ((unsigned)(bin + 1) << TAIL_POS_BITS) | (unsigned)pos; }
