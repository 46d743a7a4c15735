// Micro-benchmark: sustained fp32 VALU rate on MI355X for scalar (v_fma_f32 / v_add_f32) and packed
// (v_pk_fma_f32 / v_pk_add_f32) instructions, four waves per SIMD, independent accumulators.  Prints
// wave-instructions per ns per SIMD and GFLOP/s; the effective clock follows from the known issue cost.
// build: hipcc --offload-arch=gfx950 -O3 -o valu_rate valu_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP16(X) X X X X X X X X X X X X X X X X

template <int MODE> __global__ __launch_bounds__(256, 4) void k(float *out, int iters) {
    float a[16], b = threadIdx.x * 1e-6f + 1.0f, c = 0.999f;
#pragma unroll
    for (int i = 0; i < 16; ++i) a[i] = i * 0.01f + threadIdx.x * 1e-7f;
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 p[8], pb = {b, b}, pc = {c, c};
#pragma unroll
    for (int i = 0; i < 8; ++i) p[i] = f2{a[2 * i], a[2 * i + 1]};
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {          // v_fma_f32 x 64
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int i = 0; i < 16; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(c), "v"(b));
        } else if (MODE == 1) {   // v_add_f32 x 64
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int i = 0; i < 16; ++i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
        } else if (MODE == 2) {   // v_pk_fma_f32 x 32 (= 64 lane-FMAs per lane)
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(pc), "v"(pb));
        } else {                  // v_pk_add_f32 x 32
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(pb));
        }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += a[i];
#pragma unroll
    for (int i = 0; i < 8; ++i) s += p[i].x + p[i].y;
    if (s == 1.2345e-30f) out[0] = s;
}

template <int MODE> void run(const char *name, int ops_per_iter, int flops_per_op, float *d) {
    const int iters = 20000, blocks = 256 * 4;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, iters);
    hipEventRecord(e0);
    for (int w = 0; w < 5; ++w) hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    ms /= 5;
    const double winstr = (double)blocks * 4 * iters * ops_per_iter;          // wave-instructions
    const double per_simd_ns = winstr / 1024.0 / (ms * 1e6);
    printf("%-14s %8.3f ms  %.3f wave-instr/ns/SIMD  %.1f TFLOP/s\n", name, ms, per_simd_ns,
           winstr * 64 * flops_per_op / (ms * 1e-3) / 1e12);
}

int main() {
    float *d;
    hipMalloc(&d, 4096);
    for (int rep = 0; rep < 2; ++rep) {
        run<0>("v_fma_f32", 64, 2, d);
        run<1>("v_add_f32", 64, 1, d);
        run<2>("v_pk_fma_f32", 32, 4, d);
        run<3>("v_pk_add_f32", 32, 2, d);
    }
    return 0;
}
