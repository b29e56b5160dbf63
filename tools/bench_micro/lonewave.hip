// What does an instruction cost a wave that has its SIMD (and most of the device) to itself?  The single-sweep walk
// (k_star_walk_few) is a handful of waves on an otherwise empty MI355X.  G workgroups of one wave each run
//   dep    one dependent chain of v_mul_f32
//   walk   the walk's step: two dependent chains of three (mean) and four (deviation) instructions
//   indep  eight independent chains of v_mul_f32
// timed three ways: s_memtime (shader-clock counter) and s_memrealtime (100 MHz) inside the kernel, hipEvents outside.
//   hipcc --offload-arch=gfx950 -O3 tools/bench_micro/lonewave.hip -o tools/bench_micro/lonewave && tools/bench_micro/lonewave
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t err_ = (x); if (err_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(err_)); return 1; } } while (0)

template <int KIND>
__global__ __launch_bounds__(64) void k(unsigned iters, float seed, unsigned long long* out, float* sink)
{
    float a = seed + threadIdx.x, d = seed * 0.5f, w = 1.0001f, u = 0.9999f, s = seed;
    float c[8];
    for (int i = 0; i < 8; i++)
        c[i] = seed + i;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (unsigned it = 0; it < iters; it++) {
#pragma unroll
        for (int rep = 0; rep < 16; rep++) {
            if (KIND == 0) {
                asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a) : "v"(w));
            } else if (KIND == 1) {
                float t;
                asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a) : "v"(w));
                asm volatile("v_add_f32 %0, %0, %1" : "+v"(a) : "v"(s));
                asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a) : "v"(u));
                asm volatile("v_mul_f32 %0, %0, %1" : "+v"(d) : "v"(w));
                asm volatile("v_sub_f32 %0, %1, %2" : "=v"(t) : "v"(s), "v"(a));
                asm volatile("v_add_f32_e64 %0, %0, |%1|" : "+v"(d) : "v"(t));
                asm volatile("v_mul_f32 %0, %0, %1" : "+v"(d) : "v"(u));
            } else {
#pragma unroll
                for (int i = 0; i < 8; i++)
                    asm volatile("v_mul_f32 %0, %0, %1" : "+v"(c[i]) : "v"(w));
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) {
        out[2 * blockIdx.x] = t1 - t0;
        out[2 * blockIdx.x + 1] = r1 - r0;
    }
    float acc = a + d;
    for (int i = 0; i < 8; i++)
        acc += c[i];
    if (acc == 12345.678f)
        sink[0] = acc;
}

int main()
{
    unsigned long long* out;
    float* sink;
    CK(hipMalloc(&out, 2 * 8192 * sizeof(unsigned long long)));
    CK(hipMalloc(&sink, 4));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const unsigned iters = 2000;
    const char* names[3] = { "dep   (1 chain of v_mul_f32)", "walk  (3 + 4 instruction chains per step)", "indep (8 chains of v_mul_f32)" };
    const int per_rep[3] = { 1, 7, 8 };
    printf("G workgroups of one wave; per instruction: shader cycles (s_memtime), ns (s_memrealtime, 100 MHz), ns (hipEvents around the launch)\n");
    for (int kind = 0; kind < 3; kind++)
        for (unsigned G : { 1u, 6u, 18u, 256u, 1024u, 4096u }) {
            float ms = 0;
            for (int pass = 0; pass < 3; pass++) {
                CK(hipEventRecord(e0, 0));
                if (kind == 0)
                    hipLaunchKernelGGL(k<0>, dim3(G), dim3(64), 0, 0, iters, 1.0f, out, sink);
                else if (kind == 1)
                    hipLaunchKernelGGL(k<1>, dim3(G), dim3(64), 0, 0, iters, 1.0f, out, sink);
                else
                    hipLaunchKernelGGL(k<2>, dim3(G), dim3(64), 0, 0, iters, 1.0f, out, sink);
                CK(hipEventRecord(e1, 0));
                CK(hipEventSynchronize(e1));
                CK(hipEventElapsedTime(&ms, e0, e1));
            }
            std::vector<unsigned long long> h(2 * G);
            CK(hipMemcpy(h.data(), out, 2 * G * sizeof(unsigned long long), hipMemcpyDeviceToHost));
            double cyc = 0, rt = 0;
            for (unsigned g = 0; g < G; g++) {
                cyc += (double)h[2 * g];
                rt += (double)h[2 * g + 1];
            }
            const double n = (double)iters * 16 * per_rep[kind];
            printf("%-44s G=%5u  %6.2f cycles  %6.2f ns  %6.2f ns (launch)  -> s_memtime at %.0f MHz\n", names[kind], G, cyc / G / n, rt / G * 10.0 / n,
                   ms * 1e6 / n, (cyc / G) / (rt / G * 10.0) * 1000.0);
        }
    return 0;
}
