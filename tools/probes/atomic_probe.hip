// Probe: cost of per-workgroup column sums published as fp64 atomics (1 copy / 8 per-XCD copies) versus slab stores.
// hipcc --offload-arch=gfx950 -O3 -o atomic_probe atomic_probe.hip && ./atomic_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
__global__ __launch_bounds__(256) void k_atomic(double* sums, int fp, int copies, int spin) {
    // some fake work so that arrival is spread like a real kernel
    double a = threadIdx.x * 1e-3, b = blockIdx.x * 1e-3;
    for (int i = 0; i < spin; ++i) a = a * 1.0000001 + b;
    double* dst = sums + (size_t)(blockIdx.x % copies) * fp * 2;
    for (int c = threadIdx.x; c < fp * 2; c += blockDim.x) atomicAdd(dst + c, a + c);
}
__global__ __launch_bounds__(256) void k_atomic_f32(float* sums, int fp, int copies, int spin) {
    float a = threadIdx.x * 1e-3f, b = blockIdx.x * 1e-3f;
    for (int i = 0; i < spin; ++i) a = a * 1.0000001f + b;
    float* dst = sums + (size_t)(blockIdx.x % copies) * fp * 2;
    for (int c = threadIdx.x; c < fp * 2; c += blockDim.x) atomicAdd(dst + c, a + c);
}
__global__ __launch_bounds__(256) void k_slab(double* slab, int fp, int spin) {
    double a = threadIdx.x * 1e-3, b = blockIdx.x * 1e-3;
    for (int i = 0; i < spin; ++i) a = a * 1.0000001 + b;
    double* dst = slab + (size_t)blockIdx.x * fp * 2;
    for (int c = threadIdx.x; c < fp * 2; c += blockDim.x) dst[c] = a + c;
}
__global__ __launch_bounds__(256) void k_none(double* slab, int fp, int spin) {
    double a = threadIdx.x * 1e-3, b = blockIdx.x * 1e-3;
    for (int i = 0; i < spin; ++i) a = a * 1.0000001 + b;
    if (a == 123.456) slab[0] = a;
}
template <typename F> float timeit(F f, int n = 50) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 5; ++i) f();
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < n; ++i) f();
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms * 1000.f / n;
}
int main() {
    const int fps[] = {400, 704, 1280};
    const int wgs[] = {344, 688, 2752, 11008};
    double* buf; hipMalloc(&buf, (size_t)11008 * 1280 * 2 * 8);
    hipMemset(buf, 0, (size_t)11008 * 1280 * 2 * 8);
    for (int spin : {0, 2000}) for (int fp : fps) for (int g : wgs) {
        float t0 = timeit([&] { k_none<<<g, 256>>>(buf, fp, spin); });
        float ts = timeit([&] { k_slab<<<g, 256>>>(buf, fp, spin); });
        float t1 = timeit([&] { k_atomic<<<g, 256>>>(buf, fp, 1, spin); });
        float t8 = timeit([&] { k_atomic<<<g, 256>>>(buf, fp, 8, spin); });
        float t32 = timeit([&] { k_atomic<<<g, 256>>>(buf, fp, 32, spin); });
        float f8 = timeit([&] { k_atomic_f32<<<g, 256>>>((float*)buf, fp, 8, spin); });
        printf("spin %4d fp %4d wgs %5d: none %6.1f slab %6.1f  atomic x1 %7.1f  x8 %7.1f  x32 %7.1f  f32 x8 %7.1f us\n", spin, fp, g, t0, ts, t1, t8, t32, f8);
    }
    return 0;
}
