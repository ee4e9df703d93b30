#pragma once
// marex_tails.hip.h -- "tails": every (cell, dayofyear) histogram as short SORTED lists of its samples.
//
// The reference counts every anomaly into a dense (dayofyear, bin) histogram per cell and only ever asks it for one
// high quantile (detect.py:2638-2648, 2465-2559).  For q >= 0.6 the answer is decided by the largest samples of the
// dayofyear buckets in the pooled window; everything below only has to be COUNTED.  The device therefore keeps the
// samples of a bucket as 16-bit keys in lists that are sorted descending, so that a consumer reads the top of each list
// and stops at the first key it does not need:
//
//   key   = ((bin + 1) << 7) | pos      bin = np.digitize(anom, edges) - 1 (< nb <= 511), pos = position of the sample
//                                       inside its dayofyear bucket (output row doy_rows[doy_start[d] + pos], < 128);
//                                       0 = empty slot.  Samples the reference's histogram drops (NaN, >= edges[nb])
//                                       have no key.
//   lists[d][p][j][c] (uint4 = 8 keys)  chunk j (0: the 8 largest ... 3: the 8 smallest) of list p of bucket (dayofyear
//                                       d + 1, cell c).  A bucket's keys are PARTITIONED over its NPER =
//                                       ceil(max_bucket / 32) lists of <= 32 keys, each sorted descending (list p holds
//                                       rows 32 p .. 32 p + 31 of the bucket).  Chunk-major: one 16-byte load per lane,
//                                       contiguous across consecutive cells; consumers rarely go past chunk 0.
//   aux[d][c] (uint32)                  bits 0..9: number of keys of the bucket (= the samples the reference counts);
//                                       bit 15: the bucket holds a non-NaN value >= edges[nb] ("beyond the table": no key, the
//                                       histogram drops it, but it IS an extreme of any finite threshold -- only the mask
//                                       cares); bits 16..22 / 23..29: position of the first / second such sample of the
//                                       bucket, bit 30: the second exists, bit 31: there are more than two (only then does
//                                       the mask kernel have to look at the bucket's values)
//
// The sorting network works on PACKED PAIRS: one 32-bit register holds the keys of two independent buckets (two
// neighbouring dayofyears of one cell), v_pk_max_u16 / v_pk_min_u16 order both at once.
#include <stdint.h>

#if defined(__HIPCC__)  // hipcc: host + device; a plain host compiler (tests/host/tail_networks_check.cpp): inline
#define MAREX_HD __host__ __device__ __forceinline__
#else
#define MAREX_HD inline
#endif

#define TAIL_POS_BITS 7
#define TAIL_MAX_BUCKET 128  // pos < 2^7
#define TAIL_MAX_NB 511      // bin + 1 < 2^9
#define TAIL_LIST 32         // keys per list
#define TAIL_CH (TAIL_LIST / 8)  // 16-byte chunks per list
#define TAIL_MAX_NPER (TAIL_MAX_BUCKET / TAIL_LIST)

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

// v[0..N) bitonic -> sorted descending
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

// 16 keys, any order -> sorted descending: Batcher's odd-even merge sort, 63 compare-exchanges (the bitonic sorter needs
// 80).  The list is generated and checked exhaustively (0-1 principle, all 2^16 inputs) by tests/test_tail_networks.py.
MAREX_HD void sort16_desc(unsigned (&v)[16]) {
#define CX(a, b) pk_cx(v[a], v[b])
    CX(0, 1); CX(2, 3); CX(0, 2); CX(1, 3); CX(1, 2); CX(4, 5); CX(6, 7); CX(4, 6); CX(5, 7); CX(5, 6); CX(0, 4);
    CX(2, 6); CX(2, 4); CX(1, 5); CX(3, 7); CX(3, 5); CX(1, 2); CX(3, 4); CX(5, 6); CX(8, 9); CX(10, 11); CX(8, 10);
    CX(9, 11); CX(9, 10); CX(12, 13); CX(14, 15); CX(12, 14); CX(13, 15); CX(13, 14); CX(8, 12); CX(10, 14);
    CX(10, 12); CX(9, 13); CX(11, 15); CX(11, 13); CX(9, 10); CX(11, 12); CX(13, 14); CX(0, 8); CX(4, 12); CX(4, 8);
    CX(2, 10); CX(6, 14); CX(6, 10); CX(2, 4); CX(6, 8); CX(10, 12); CX(1, 9); CX(5, 13); CX(5, 9); CX(3, 11);
    CX(7, 15); CX(7, 11); CX(3, 5); CX(7, 9); CX(11, 13); CX(1, 2); CX(3, 4); CX(5, 6); CX(7, 8); CX(9, 10);
    CX(11, 12); CX(13, 14);
#undef CX
}

MAREX_HD unsigned tail_key(int bin, int pos) { return ((unsigned)(bin + 1) << TAIL_POS_BITS) | (unsigned)pos; }

// Samples beyond the table, per bucket, while the producer walks it in time order: a 16-bit state [n:2][pos1:7][pos2:7]
// (n = 3: more than two), turned into the upper half of the aux word at the end.
#define TAIL_AUX_COUNT 0x3FFu
#define TAIL_AUX_BEYOND 0x8000u     // at least one sample beyond the table
#define TAIL_AUX_SECOND 0x40000000u // pos2 is valid
#define TAIL_AUX_MANY 0x80000000u   // more than two: look at the values
MAREX_HD unsigned tail_ovf_add(unsigned st, unsigned pos) {
    const unsigned n = st & 3u;
    return n == 0u ? (1u | (pos << 2)) : (n == 1u ? (2u | (st & 0x1FCu) | (pos << 9)) : (st | 3u));
}
MAREX_HD unsigned tail_aux_word(unsigned cnt, unsigned st) {
    const unsigned n = st & 3u;
    return cnt | (n ? TAIL_AUX_BEYOND : 0u) | (((st >> 2) & 0x7Fu) << 16) | (((st >> 9) & 0x7Fu) << 23) |
           (n >= 2u ? TAIL_AUX_SECOND : 0u) | (n == 3u ? TAIL_AUX_MANY : 0u);
}
