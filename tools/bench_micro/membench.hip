// Streaming ceiling for k_split's traffic shape: per point 12 B read (x, y, z) and 28 B written
// (four f32 + one u16 ring-sorted, two f32 + one u16 sector-sorted), 2048-point tiles, 512 threads.
//   hipcc --offload-arch=gfx950 -O3 tools/bench_micro/membench.hip -o tools/bench_micro/membench && ./membench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CK(x) do { hipError_t err_ = (x); if (err_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(err_)); return 1; } } while (0)
template <int MODE>
__global__ __launch_bounds__(512) void k(const float* __restrict__ x, const float* __restrict__ y, const float* __restrict__ z,
                                         float* a, float* b, float* c, float* d, uint16_t* e, float* f, float* g, uint16_t* h, size_t n)
{
    const size_t base = (size_t)blockIdx.x * 2048;
    float px[4], py[4], pz[4];
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const size_t i = base + q * 512 + threadIdx.x;
        px[q] = x[i]; py[q] = y[i]; pz[q] = z[i];
    }
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const size_t i = base + q * 512 + threadIdx.x;
        const float s = px[q] + py[q], t = pz[q] * 2.f;
        if (MODE == 0) {
            __builtin_nontemporal_store(px[q], &a[i]); __builtin_nontemporal_store(py[q], &b[i]);
            __builtin_nontemporal_store(pz[q], &c[i]); __builtin_nontemporal_store(s, &d[i]);
            __builtin_nontemporal_store((uint16_t)q, &e[i]);
            __builtin_nontemporal_store(t, &f[i]); __builtin_nontemporal_store(pz[q], &g[i]);
            __builtin_nontemporal_store((uint16_t)(q + 1), &h[i]);
        } else if (MODE == 1) {
            a[i] = px[q]; b[i] = py[q]; c[i] = pz[q]; d[i] = s; e[i] = (uint16_t)q; f[i] = t; g[i] = pz[q]; h[i] = (uint16_t)(q + 1);
        } else if (MODE == 2) {   // read only
            if (s + t == 12345.678f) a[i] = s;
        } else if (MODE == 3) {   // 24 B written: no u16 arrays, d dropped
            __builtin_nontemporal_store(px[q], &a[i]); __builtin_nontemporal_store(py[q], &b[i]);
            __builtin_nontemporal_store(pz[q], &c[i]);
            __builtin_nontemporal_store(t, &f[i]); __builtin_nontemporal_store(pz[q], &g[i]);
            __builtin_nontemporal_store(s, &d[i]);
        }
    }
}
int main()
{
    const size_t n = (size_t)1024 * 131072;
    float *x, *y, *z, *a, *b, *c, *d, *f, *g; uint16_t *e, *h;
    CK(hipMalloc(&x, n * 4)); CK(hipMalloc(&y, n * 4)); CK(hipMalloc(&z, n * 4));
    CK(hipMalloc(&a, n * 4)); CK(hipMalloc(&b, n * 4)); CK(hipMalloc(&c, n * 4)); CK(hipMalloc(&d, n * 4));
    CK(hipMalloc(&f, n * 4)); CK(hipMalloc(&g, n * 4)); CK(hipMalloc(&e, n * 2)); CK(hipMalloc(&h, n * 2));
    CK(hipMemset(x, 0, n * 4)); CK(hipMemset(y, 0, n * 4)); CK(hipMemset(z, 0, n * 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const char* names[4] = { "12R+28W nontemporal", "12R+28W plain", "12R only", "12R+24W nontemporal" };
    const double bytes[4] = { 40.0, 40.0, 12.0, 36.0 };
    for (int mode = 0; mode < 4; mode++) {
        float best = 1e9f;
        for (int it = 0; it < 6; it++) {
            CK(hipEventRecord(e0));
            const dim3 grid(n / 2048), blk(512);
            if (mode == 0) hipLaunchKernelGGL(k<0>, grid, blk, 0, 0, x, y, z, a, b, c, d, e, f, g, h, n);
            if (mode == 1) hipLaunchKernelGGL(k<1>, grid, blk, 0, 0, x, y, z, a, b, c, d, e, f, g, h, n);
            if (mode == 2) hipLaunchKernelGGL(k<2>, grid, blk, 0, 0, x, y, z, a, b, c, d, e, f, g, h, n);
            if (mode == 3) hipLaunchKernelGGL(k<3>, grid, blk, 0, 0, x, y, z, a, b, c, d, e, f, g, h, n);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (it && ms < best) best = ms;
        }
        printf("%-22s %.3f ms  %.2f TB/s\n", names[mode], best, bytes[mode] * n / best * 1e-9);
    }
    return 0;
}
