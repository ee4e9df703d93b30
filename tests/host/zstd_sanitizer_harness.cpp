#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <cstring>
#include <vector>
#include <string>
#include <dirent.h>
extern "C" int marex_zstd_decompress_h(const uint8_t* src, int64_t n, uint8_t* dst, int64_t cap, int64_t* out_n);
int main() {
    DIR* d = opendir(".");
    int n = 0, ok = 0;
    while (dirent* e = readdir(d)) {
        std::string name = e->d_name;
        if (name.size() < 5 || name.substr(name.size() - 4) != ".bin") continue;
        const size_t us = name.find('_');
        const long cap = atol(name.substr(us + 1).c_str());
        FILE* f = fopen(name.c_str(), "rb");
        fseek(f, 0, SEEK_END); long len = ftell(f); fseek(f, 0, SEEK_SET);
        // exact-size heap buffers so that the sanitizer sees any access past either end
        uint8_t* src = (uint8_t*)malloc(len ? len : 1);
        if (len) fread(src, 1, len, f);
        fclose(f);
        uint8_t* dst = (uint8_t*)malloc(cap ? cap : 1);
        int64_t got = 0;
        int rc = marex_zstd_decompress_h(src, len, dst, cap, &got);
        ok += rc == 0; ++n;
        free(src); free(dst);
    }
    printf("%d streams, %d decoded, %d refused\n", n, ok, n - ok);
    return 0;
}
