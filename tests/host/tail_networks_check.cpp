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

static void check_sort16(std::mt19937& rng, int range) {
    unsigned v[16];
    std::vector<unsigned short> lo(16), hi(16);
    for (int i = 0; i < 16; ++i) {
        lo[i] = (unsigned short)(rng() % range);
        hi[i] = (unsigned short)(rng() % range);
        v[i] = lo[i] | ((unsigned)hi[i] << 16);
    }
    sort16_desc(v);
    std::sort(lo.begin(), lo.end(), std::greater<unsigned short>());
    std::sort(hi.begin(), hi.end(), std::greater<unsigned short>());
    for (int i = 0; i < 16; ++i) CHECK(v[i] == (lo[i] | ((unsigned)hi[i] << 16)));
}

// 0-1 principle: a comparator network that sorts every 0/1 input sorts everything -- all 2^16 inputs, both halves
static void check_sort16_exhaustive() {
    for (unsigned bits = 0; bits < 65536u; ++bits) {
        unsigned v[16];
        for (int i = 0; i < 16; ++i) v[i] = ((bits >> i) & 1u) | ((((~bits) >> i) & 1u) << 16);
        sort16_desc(v);
        for (int i = 0; i + 1 < 16; ++i) CHECK((v[i] & 0xFFFFu) >= (v[i + 1] & 0xFFFFu) && (v[i] >> 16) >= (v[i + 1] >> 16));
    }
}

int main() {
    std::mt19937 rng(12345);
    check_sort16_exhaustive();
    for (int it = 0; it < 2000; ++it) {
        const int range = (it % 3 == 0) ? 4 : ((it % 3 == 1) ? 600 : 65536);  // many ties / typical keys / full range
        check_sort<8>(rng, range);
        check_sort<16>(rng, range);
        check_sort<32>(rng, range);
        check_sort16(rng, range);
    }
    CHECK(tail_key(0, 0) == 128u && tail_key(501, 127) == ((502u << 7) | 127u));
    printf(fails ? "FAILED %d\n" : "OK\n", fails);
    return fails ? 1 : 0;
}
