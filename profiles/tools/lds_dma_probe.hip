// lds_dma_probe.hip -- pins the addressing of `buffer_load_dword / dwordx4 ... offen lds` (LDS-DMA) on gfx950 before the
// anomaly kernel relies on it: which LDS bytes a wave-instruction writes (M0 base, inst_offset, lane order), which memory
// bytes it reads (descriptor base + soffset + inst_offset + per-lane voffset), what an out-of-range lane does with a
// bounded descriptor, and that `s_waitcnt vmcnt(N)` counts these loads in issue order.
// Build: hipcc --offload-arch=gfx950 -O2 -o lds_dma_probe lds_dma_probe.hip ; run on the GPU box: ./lds_dma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef int rsrc_t __attribute__((ext_vector_type(4)));

__device__ __forceinline__ rsrc_t make_rsrc(const void* base, unsigned num_records) {
    const unsigned long long b = (unsigned long long)base;
    rsrc_t r;
    r.x = __builtin_amdgcn_readfirstlane((int)(unsigned)b);
    r.y = __builtin_amdgcn_readfirstlane((int)((unsigned)(b >> 32) & 0xFFFFu));
    r.z = __builtin_amdgcn_readfirstlane((int)num_records);
    r.w = 0x00020000;
    return r;
}

#define NLDS 2048  // dwords

// mode 0: dword, inst offset 256, soffset 1024, M0 = base + 512
// mode 1: dwordx4, per-lane voffset = (lane >> 4) * rowb + (lane & 15) * 16, inst offset 0, M0 = base + 1024
// mode 2: dwordx4 with inst offset 1024 and soffset, M0 = base
// mode 3: dword, bounded descriptor (num_records = 128 bytes): lanes >= 32 are out of range
__global__ void k_probe(const float* src, float* out, int mode, int rowb) {
    __shared__ float lds[NLDS];
    const int lane = threadIdx.x;
    for (int i = lane; i < NLDS; i += 64) lds[i] = -1.0f;
    __syncthreads();
    typedef __attribute__((address_space(3))) float lds_f;
    const unsigned lbase = (unsigned)(size_t)(lds_f*)lds;
    unsigned keep;
    if (mode == 0) {
        const rsrc_t r = make_rsrc(src, 0xFFFFFFFFu);
        const unsigned voff = lane * 4, m0v = __builtin_amdgcn_readfirstlane(lbase + 512), soff = __builtin_amdgcn_readfirstlane(1024);
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dword %1, %2, %4 offen offset:256 lds\n\ts_mov_b32 m0, %0\n\ts_waitcnt vmcnt(0)"
                     : "=&s"(keep) : "v"(voff), "s"(r), "s"(m0v), "s"(soff) : "memory");
    } else if (mode == 1) {
        const rsrc_t r = make_rsrc(src, 0xFFFFFFFFu);
        const unsigned voff = (lane >> 4) * rowb + (lane & 15) * 16, m0v = __builtin_amdgcn_readfirstlane(lbase + 1024), soff = __builtin_amdgcn_readfirstlane(0);
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %4 offen lds\n\ts_mov_b32 m0, %0\n\ts_waitcnt vmcnt(0)"
                     : "=&s"(keep) : "v"(voff), "s"(r), "s"(m0v), "s"(soff) : "memory");
    } else if (mode == 2) {
        const rsrc_t r = make_rsrc(src, 0xFFFFFFFFu);
        const unsigned voff = (lane >> 4) * rowb + (lane & 15) * 16, m0v = __builtin_amdgcn_readfirstlane(lbase), soff = __builtin_amdgcn_readfirstlane(2048);
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %4 offen offset:1024 lds\n\ts_mov_b32 m0, %0\n\ts_waitcnt vmcnt(0)"
                     : "=&s"(keep) : "v"(voff), "s"(r), "s"(m0v), "s"(soff) : "memory");
    } else {
        const rsrc_t r = make_rsrc(src, 128u);
        const unsigned voff = lane * 4, m0v = __builtin_amdgcn_readfirstlane(lbase), soff = __builtin_amdgcn_readfirstlane(0);
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dword %1, %2, %4 offen lds\n\ts_mov_b32 m0, %0\n\ts_waitcnt vmcnt(0)"
                     : "=&s"(keep) : "v"(voff), "s"(r), "s"(m0v), "s"(soff) : "memory");
    }
    __syncthreads();
    for (int i = lane; i < NLDS; i += 64) out[i] = lds[i];
}

int main() {
    const int N = 1 << 20, rowb = 40000;  // bytes per "row" of the x4 probe
    std::vector<float> h(N);
    for (int i = 0; i < N; ++i) h[i] = (float)i;
    float *src, *out;
    if (hipMalloc(&src, N * 4) != hipSuccess || hipMalloc(&out, NLDS * 4) != hipSuccess) return 1;
    hipMemcpy(src, h.data(), N * 4, hipMemcpyHostToDevice);
    std::vector<float> o(NLDS);
    int bad = 0;
    for (int mode = 0; mode < 4; ++mode) {
        hipLaunchKernelGGL(k_probe, dim3(1), dim3(64), 0, 0, src, out, mode, rowb);
        if (hipDeviceSynchronize() != hipSuccess) { printf("mode %d: launch failed\n", mode); return 2; }
        hipMemcpy(o.data(), out, NLDS * 4, hipMemcpyDeviceToHost);
        // expectation under the documented rule: LDS byte = M0 + inst_offset + lane * size; memory byte = base + soffset + inst_offset + voffset
        std::vector<float> e(NLDS, -1.0f);
        if (mode == 0) for (int l = 0; l < 64; ++l) e[(512 + 256) / 4 + l] = (float)((1024 + 256) / 4 + l);
        if (mode == 1) for (int l = 0; l < 64; ++l) for (int k = 0; k < 4; ++k) e[1024 / 4 + l * 4 + k] = (float)(((l >> 4) * rowb + (l & 15) * 16) / 4 + k);
        if (mode == 2) for (int l = 0; l < 64; ++l) for (int k = 0; k < 4; ++k) e[1024 / 4 + l * 4 + k] = (float)((2048 + 1024 + (l >> 4) * rowb + (l & 15) * 16) / 4 + k);
        if (mode == 3) for (int l = 0; l < 64; ++l) e[l] = l < 32 ? (float)l : 0.0f;
        int nb = 0, first = -1;
        for (int i = 0; i < NLDS; ++i) if (o[i] != e[i]) { if (first < 0) first = i; ++nb; }
        printf("mode %d: %s", mode, nb ? "MISMATCH" : "as expected");
        if (nb) {
            printf(" (%d dwords, first at %d: got %g, expected %g); written dwords:", nb, first, o[first], e[first]);
            int shown = 0;
            for (int i = 0; i < NLDS && shown < 12; ++i) if (o[i] != -1.0f) { printf(" [%d]=%g", i, o[i]); ++shown; }
        }
        printf("\n");
        bad += nb != 0;
    }
    return bad ? 3 : 0;
}
