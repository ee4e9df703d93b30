// valu_rates.hip -- issue-rate microbenchmark for the vector instructions the hot-path kernels lean on (gfx950).
// Build: hipcc --offload-arch=gfx950 -O3 -o valu_rates valu_rates.hip ; run on the GPU box: ./valu_rates
// Prints wave-instructions per cycle per SIMD at 1, 2, 4 resident waves per SIMD (cycle = s_memtime tick).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP 64
#define ITER 2000

#define BODY8(OP)                                                                                                    \
    asm volatile(OP "\n" : "+v"(a0) : "v"(b));                                                                       \
    asm volatile(OP "\n" : "+v"(a1) : "v"(b));                                                                       \
    asm volatile(OP "\n" : "+v"(a2) : "v"(b));                                                                       \
    asm volatile(OP "\n" : "+v"(a3) : "v"(b));                                                                       \
    asm volatile(OP "\n" : "+v"(a4) : "v"(b));                                                                       \
    asm volatile(OP "\n" : "+v"(a5) : "v"(b));                                                                       \
    asm volatile(OP "\n" : "+v"(a6) : "v"(b));                                                                       \
    asm volatile(OP "\n" : "+v"(a7) : "v"(b));

template <int KIND>
__global__ void __launch_bounds__(256) k_rate(unsigned long long* cyc, float* sink) {
    typedef float v2f __attribute__((ext_vector_type(2)));
    const float seed = (float)threadIdx.x * 1e-3f;
    unsigned long long t0 = 0, t1 = 0;
    if (KIND == 0 || KIND == 1) {
        float a0 = seed, a1 = seed + 1, a2 = seed + 2, a3 = seed + 3, a4 = seed + 4, a5 = seed + 5, a6 = seed + 6, a7 = seed + 7;
        float b = 1.000001f;
        t0 = __builtin_amdgcn_s_memtime();
        for (int it = 0; it < ITER; ++it) {
#pragma unroll
            for (int r = 0; r < REP / 8; ++r) {
                if (KIND == 0) { BODY8("v_add_f32 %0, %0, %1") } else { BODY8("v_fma_f32 %0, %0, %1, %1") }
            }
        }
        t1 = __builtin_amdgcn_s_memtime();
        sink[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    } else if (KIND == 2 || KIND == 3) {
        v2f a0 = {seed, seed}, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, a4 = a0 + 4.f, a5 = a0 + 5.f, a6 = a0 + 6.f, a7 = a0 + 7.f;
        v2f b = {1.000001f, 0.999999f};
        t0 = __builtin_amdgcn_s_memtime();
        for (int it = 0; it < ITER; ++it) {
#pragma unroll
            for (int r = 0; r < REP / 8; ++r) {
                if (KIND == 2) { BODY8("v_pk_add_f32 %0, %0, %1") } else { BODY8("v_pk_fma_f32 %0, %0, %1, %1") }
            }
        }
        t1 = __builtin_amdgcn_s_memtime();
        sink[blockIdx.x * 256 + threadIdx.x] = a0.x + a1.y + a2.x + a3.y + a4.x + a5.y + a6.x + a7.y;
    } else if (KIND == 8 || KIND == 9) {
        v2f a0 = {seed, seed}, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, a4 = a0 + 4.f, a5 = a0 + 5.f, a6 = a0 + 6.f, a7 = a0 + 7.f;
        v2f b = {1.000001f, 0.999999f};
        t0 = __builtin_amdgcn_s_memtime();
        for (int it = 0; it < ITER; ++it) {
#pragma unroll
            for (int r = 0; r < REP / 8; ++r) {
                if (KIND == 8) { BODY8("v_pk_mov_b32 %0, %1, %1 op_sel:[0,1]") } else { BODY8("v_mov_b64 %0, %1") }
            }
        }
        t1 = __builtin_amdgcn_s_memtime();
        sink[blockIdx.x * 256 + threadIdx.x] = a0.x + a1.y + a2.x + a3.y + a4.x + a5.y + a6.x + a7.y;
    } else {
        unsigned a0 = threadIdx.x, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 * 11, a5 = a0 * 13, a6 = a0 * 17, a7 = a0 * 19;
        unsigned b = 0x01230123u + threadIdx.x;
        t0 = __builtin_amdgcn_s_memtime();
        for (int it = 0; it < ITER; ++it) {
#pragma unroll
            for (int r = 0; r < REP / 8; ++r) {
                if (KIND == 4) { BODY8("v_pk_max_u16 %0, %0, %1") }
                else if (KIND == 5) { BODY8("v_max_u32 %0, %0, %1") }
                else if (KIND == 6) { BODY8("v_pk_add_u16 %0, %0, %1") }
                else { BODY8("v_mad_u32_u24 %0, %0, %1, %1") }
            }
        }
        t1 = __builtin_amdgcn_s_memtime();
        sink[blockIdx.x * 256 + threadIdx.x] = (float)(a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7);
    }
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

// Do instruction TYPES of different waves of one SIMD issue in the same cycle?  Half of the workgroups (roles mixed on every
// CU: role = parity of popcount(blockIdx >> 3)) run a chain of role-A instructions, the other half role-B instructions;
// A / B = 0 nothing (exit), 1 v_fma_f32, 2 s_add_u32, 3 ds_read_b32 (same address, no conflicts), 4 s_nop 0.
#define SBODY8(OP)                                                                                                   \
    asm volatile(OP "\n" : "+s"(s0) : "s"(sb));                                                                      \
    asm volatile(OP "\n" : "+s"(s1) : "s"(sb));                                                                      \
    asm volatile(OP "\n" : "+s"(s2) : "s"(sb));                                                                      \
    asm volatile(OP "\n" : "+s"(s3) : "s"(sb));                                                                      \
    asm volatile(OP "\n" : "+s"(s4) : "s"(sb));                                                                      \
    asm volatile(OP "\n" : "+s"(s5) : "s"(sb));                                                                      \
    asm volatile(OP "\n" : "+s"(s6) : "s"(sb));                                                                      \
    asm volatile(OP "\n" : "+s"(s7) : "s"(sb));

__device__ __forceinline__ float chain(int kind, float seed, float* lds) {
    if (kind == 1) {
        float a0 = seed, a1 = seed + 1, a2 = seed + 2, a3 = seed + 3, a4 = seed + 4, a5 = seed + 5, a6 = seed + 6, a7 = seed + 7;
        float b = 1.000001f;
        for (int it = 0; it < ITER; ++it) {
#pragma unroll
            for (int r = 0; r < REP / 8; ++r) { BODY8("v_fma_f32 %0, %0, %1, %1") }
        }
        return a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    } else if (kind == 2) {
        for (int it = 0; it < ITER; ++it) {
#pragma unroll
            for (int r = 0; r < REP / 8; ++r)
                asm volatile("s_add_u32 s20, s20, 3\n s_add_u32 s21, s21, 3\n s_add_u32 s22, s22, 3\n s_add_u32 s23, s23, 3\n"
                             "s_add_u32 s24, s24, 3\n s_add_u32 s25, s25, 3\n s_add_u32 s26, s26, 3\n s_add_u32 s27, s27, 3\n"
                             ::: "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27", "scc");
        }
        return seed;
    } else if (kind == 3) {
        float acc = 0.f;
        const unsigned addr = (threadIdx.x & 63) * 4;
        for (int it = 0; it < ITER; ++it) {
            float v0, v1, v2, v3, v4, v5, v6, v7;
#pragma unroll
            for (int r = 0; r < REP / 8; ++r) {
                asm volatile("ds_read_b32 %0, %8\n ds_read_b32 %1, %8 offset:256\n ds_read_b32 %2, %8 offset:512\n ds_read_b32 %3, %8 offset:768\n"
                             "ds_read_b32 %4, %8 offset:1024\n ds_read_b32 %5, %8 offset:1280\n ds_read_b32 %6, %8 offset:1536\n ds_read_b32 %7, %8 offset:1792\n"
                             "s_waitcnt lgkmcnt(0)\n"
                             : "=v"(v0), "=v"(v1), "=v"(v2), "=v"(v3), "=v"(v4), "=v"(v5), "=v"(v6), "=v"(v7) : "v"(addr) : "memory");
            }
            acc += v0 + v7;
        }
        return acc + lds[0];
    } else if (kind == 4) {
        for (int it = 0; it < ITER; ++it) {
#pragma unroll
            for (int r = 0; r < REP; ++r) asm volatile("s_nop 0");
        }
        return seed;
    }
    return 0.f;
}

__global__ void __launch_bounds__(256) k_mixed(int kindA, int kindB, float* sink) {
    __shared__ float lds[1024];
    for (int i = threadIdx.x; i < 1024; i += 256) lds[i] = (float)i;
    __syncthreads();
    const int role = __builtin_popcount(blockIdx.x >> 3) & 1;
    const int kind = __builtin_amdgcn_readfirstlane(role ? kindB : kindA);
    const float v = chain(kind, (float)(threadIdx.x >> 6), lds);
    sink[blockIdx.x * 256 + threadIdx.x] = v;
}

static void run_mixed(const char* name, int kindA, int kindB) {
    const int blocks = 256 * 4;  // 4 workgroups of 4 waves per CU = 4 waves per SIMD, two of each role
    float* sink;
    hipMalloc(&sink, blocks * 256 * sizeof(float));
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    k_mixed<<<blocks, 256>>>(kindA, kindB, sink);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k_mixed<<<blocks, 256>>>(kindA, kindB, sink);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    printf("mixed %-28s: %.3f ms\n", name, ms);
    hipFree(sink);
}

template <int KIND>
static void run(const char* name) {
    int dev_cus = 256;
    for (int wps : {1, 2, 4, 8}) {
        const int blocks = dev_cus * wps;  // 256-thread block = one wave per SIMD; wps blocks per CU
        unsigned long long* cyc;
        float* sink;
        hipMalloc(&cyc, blocks * 4 * sizeof(unsigned long long));
        hipMalloc(&sink, blocks * 256 * sizeof(float));
        hipEvent_t e0, e1;
        hipEventCreate(&e0);
        hipEventCreate(&e1);
        k_rate<KIND><<<blocks, 256>>>(cyc, sink);  // warm-up
        hipDeviceSynchronize();
        hipEventRecord(e0);
        k_rate<KIND><<<blocks, 256>>>(cyc, sink);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        std::vector<unsigned long long> h(blocks * 4);
        hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
        double avg = 0;
        for (auto v : h) avg += (double)v;
        avg /= h.size();
        const double ninst = (double)REP * ITER;
        // per-wave cycles per instruction (memtime ticks at 100 MHz on some parts: report both views)
        const double total_wave_inst = ninst * blocks * 4;
        printf("%-16s waves/SIMD %d: %.3f ms, %.1f G wave-inst/s chip, %.3f wave-inst/ns/SIMD, memtime ticks/inst/wave %.3f\n", name,
               wps, ms, total_wave_inst / (ms * 1e-3) / 1e9, total_wave_inst / (ms * 1e6) / 1024.0, avg / ninst);
        hipFree(cyc);
        hipFree(sink);
    }
}

int main() {
    run_mixed("fma | nothing", 1, 0);
    run_mixed("nothing | s_add", 0, 2);
    run_mixed("fma | s_add", 1, 2);
    run_mixed("fma | fma", 1, 1);
    run_mixed("s_add | s_add", 2, 2);
    run_mixed("ds_read | nothing", 3, 0);
    run_mixed("fma | ds_read", 1, 3);
    run_mixed("s_add | ds_read", 2, 3);
    run_mixed("s_nop | nothing", 4, 0);
    run_mixed("fma | s_nop", 1, 4);
    run<0>("v_add_f32");
    run<1>("v_fma_f32");
    run<2>("v_pk_add_f32");
    run<3>("v_pk_fma_f32");
    run<4>("v_pk_max_u16");
    run<5>("v_max_u32");
    run<6>("v_pk_add_u16");
    run<7>("v_mad_u32_u24");
    run<8>("v_pk_mov_b32");
    run<9>("v_mov_b64");
    return 0;
}
