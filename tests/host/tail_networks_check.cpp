// Host check of the packed sorting / merging networks of marex_amd/csrc/marex_tails.hip.h (they are plain
// __host__ __device__ code): random packed keys, compared with std::sort on each 16-bit half.
// Built and run by tests/test_tail_networks.py with the clang++ of the ROCm toolchain (no GPU involved).
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <random>
#include <vector>

#include "../../marex_amd/csrc/marex_tails.hip.h"

static int fails = 0;
#define CHECK(c)                                                  \
    do {                                                          \
        if (!(c)) {                                               \
            if (fails < 10) printf("FAIL %s:%d %s\n", __FILE__, __LINE__, #c); \
            ++fails;                                              \
        }                                                         \
    } while (0)

template <int N>
static void check_sort(std::mt19937& rng, int range) {
    unsigned v[N];
    std::vector<unsigned short> lo(N), hi(N);
    for (int i = 0; i < N; ++i) {
        lo[i] = (unsigned short)(rng() % range);
        hi[i] = (unsigned short)(rng() % range);
        v[i] = lo[i] | ((unsigned)hi[i] << 16);
    }
    bitonic_sort_desc<N>(v);
    std::sort(lo.begin(), lo.end(), std::greater<unsigned short>());
    std::sort(hi.begin(), hi.end(), std::greater<unsigned short>());
    for (int i = 0; i < N; ++i) CHECK(v[i] == (lo[i] | ((unsigned)hi[i] << 16)));
}

template <int K, int B>
static void check_merge(std::mt19937& rng, int range, int rounds) {
    unsigned top[K];
    for (int i = 0; i < K; ++i) top[i] = 0;
    std::vector<unsigned short> all_lo, all_hi;
    for (int r = 0; r < rounds; ++r) {
        unsigned nw[B];
        for (int i = 0; i < B; ++i) {
            const unsigned short a = (unsigned short)(rng() % range), b = (unsigned short)(rng() % range);
            all_lo.push_back(a);
            all_hi.push_back(b);
            nw[i] = a | ((unsigned)b << 16);
        }
        bitonic_sort_desc<B>(nw);
        tail_merge<K, B>(top, nw);
        std::vector<unsigned short> slo(all_lo), shi(all_hi);
        std::sort(slo.begin(), slo.end(), std::greater<unsigned short>());
        std::sort(shi.begin(), shi.end(), std::greater<unsigned short>());
        for (int i = 0; i < K; ++i) {
            const unsigned short el = i < (int)slo.size() ? slo[i] : 0, eh = i < (int)shi.size() ? shi[i] : 0;
            CHECK(top[i] == (el | ((unsigned)eh << 16)));
        }
    }
}

int main() {
    std::mt19937 rng(12345);
    for (int it = 0; it < 2000; ++it) {
        const int range = (it % 3 == 0) ? 4 : ((it % 3 == 1) ? 600 : 65536);  // many ties / typical keys / full range
        check_sort<8>(rng, range);
        check_sort<16>(rng, range);
        check_sort<32>(rng, range);
        check_merge<16, 16>(rng, range, 7);
        check_merge<32, 16>(rng, range, 7);
        check_merge<32, 8>(rng, range, 9);
        check_merge<16, 8>(rng, range, 9);
    }
    CHECK(tail_key(0, 0) == 128u && tail_key(501, 127) == ((502u << 7) | 127u));
    printf(fails ? "FAILED %d\n" : "OK\n", fails);
    return fails ? 1 : 0;
}
