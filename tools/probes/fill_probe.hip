// How fast can a CU fill its LDS from L2-resident data?  (design input of csrc/gemm_bx3.hip)
//   mode 0: LDS-DMA (buffer_load_dwordx4 ... lds), 1 KB per wave-instruction
//   mode 1: global_load_dwordx4 -> VGPR -> ds_write_b128 (register staging), 8 loads in flight per wave
// NW waves per workgroup, one workgroup per CU, every workgroup streams the SAME `region` bytes (L2 / MALL resident) `iters` times.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(2); } } while (0)

template <int MODE>
__global__ __launch_bounds__(512) void fill(const unsigned char* src, unsigned region, int iters, int row_bytes, unsigned* sink) {
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), nw = blockDim.x >> 6;
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, (int)region, 0x00020000);
    // a "piece" = 1 KB: row_bytes == 64: 16 rows x 64 B at stride 800 B (the NT image); row_bytes == 1024: contiguous
    unsigned acc = 0;
    const unsigned pieces = region / 1024;
    unsigned pc = (blockIdx.x * 7 + wave) % pieces;
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const unsigned off = row_bytes == 64 ? ((pc * 16 + (lane >> 2)) * 800u + (lane & 3) * 16u) % (region - 64) : pc * 1024u + lane * 16u;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(smem + (wave * 8 + u) * 1024), 16, (int)(off & ~15u), 0, 0, 0);
                pc += nw; if (pc >= pieces) pc -= pieces;
            }
            asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        } else {
            u32x4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const unsigned off = row_bytes == 64 ? ((pc * 16 + (lane >> 2)) * 800u + (lane & 3) * 16u) % (region - 64) : pc * 1024u + lane * 16u;
                v[u] = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(off & ~15u), 0, 0);
                pc += nw; if (pc >= pieces) pc -= pieces;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) *reinterpret_cast<u32x4*>(smem + (wave * 8 + u) * 1024 + lane * 16) = v[u];
        }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __syncthreads();
    acc = *reinterpret_cast<unsigned*>(smem + threadIdx.x * 4);
    if (acc == 0x12345678u) sink[0] = acc;
}

int main() {
    const unsigned region = 2u << 20;
    unsigned char* src; unsigned* sink;
    CK(hipMalloc(&src, region + 4096)); CK(hipMemset(src, 1, region + 4096)); CK(hipMalloc(&sink, 64));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&fill<0>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&fill<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
    const int iters = 2000;
    for (int mode = 0; mode < 2; ++mode)
        for (int rb : {64, 1024})
            for (int nw : {1, 2, 4, 8}) {
                for (int rep = 0; rep < 2; ++rep) {
                    CK(hipEventRecord(a, nullptr));
                    if (mode == 0) fill<0><<<256, nw * 64, 65536, nullptr>>>(src, region, iters, rb, sink);
                    else fill<1><<<256, nw * 64, 65536, nullptr>>>(src, region, iters, rb, sink);
                    CK(hipEventRecord(b, nullptr)); CK(hipEventSynchronize(b));
                    float ms; CK(hipEventElapsedTime(&ms, a, b));
                    const double bytes = 256.0 * nw * iters * 8 * 1024;
                    if (rep == 1) printf("mode %d (%s)  rows of %4d B  %d waves/CU: %7.2f TB/s chip = %6.1f GB/s per CU = %5.1f B/clk/CU @2.1GHz\n", mode,
                                         mode ? "load + ds_write" : "LDS-DMA", rb, nw, bytes / ms * 1e-9, bytes / ms * 1e-6 / 256, bytes / ms * 1e-6 / 256 / 2.1);
                }
            }
    return 0;
}
