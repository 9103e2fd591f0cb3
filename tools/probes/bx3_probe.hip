// Hardware-semantics probe for csrc/gemm_bx3.hip (run once on an MI355X: `hipcc --offload-arch=gfx950 -O2 -o bx3_probe
// bx3_probe.hip && ./bx3_probe`).  Three facts the kernel is built on, each checked against a host computation:
//   1. ds_read_b64_tr_b16 (__builtin_amdgcn_ds_read_tr16_b64): which (source lane, element) lands in which (lane, element);
//   2. the operand / result register layout of v_mfma_f32_32x32x16_bf16;
//   3. global_load_lds_dwordx4: LDS destination = wave-uniform base + lane * 16, per-lane global source.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <math.h>
#include <string.h>

typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__global__ void tr_probe(uint16_t* out) {
    __shared__ __attribute__((aligned(16))) uint16_t lds[64 * 32];
    const int lane = threadIdx.x;
    for (int i = lane; i < 64 * 32; i += 64) lds[i] = 0xFFFF;
    __syncthreads();
    for (int e = 0; e < 4; ++e) lds[lane * 32 + e] = lane * 4 + e;      // lane L owns 4 elements at byte address 64 L
    __syncthreads();
    s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(lds + lane * 32));
    for (int e = 0; e < 4; ++e) out[lane * 4 + e] = (uint16_t)v[e];
}

static inline uint16_t f2bf(float f) { union { float f; uint32_t u; } c; c.f = f; return (uint16_t)(c.u >> 16); }
static inline float bf2f(uint16_t h) { union { float f; uint32_t u; } c; c.u = (uint32_t)h << 16; return c.f; }

// C[32][32] = A[32][16] . B[32][16]^T with the ASSUMED layout: A lane (i = lane & 31, kg = lane >> 5) holds A[i][8 kg .. 8 kg + 7],
// B likewise with its row = output column; D: col = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)
__global__ void mfma_probe(const uint16_t* A, const uint16_t* B, float* C) {
    const int lane = threadIdx.x;
    const int i = lane & 31, kg = lane >> 5;
    u32x4 a = *reinterpret_cast<const u32x4*>(A + i * 16 + 8 * kg);
    u32x4 b = *reinterpret_cast<const u32x4*>(B + i * 16 + 8 * kg);
    f32x16 acc;
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc, 0, 0, 0);
    for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5), col = lane & 31;
        C[row * 32 + col] = acc[r];
    }
}

// every lane copies 16 bytes from its OWN global address (a permutation of the rows) to LDS base + lane * 16
__global__ void glds_probe(const uint32_t* src, uint32_t* out) {
    __shared__ __attribute__((aligned(16))) uint32_t lds[64 * 4];
    const int lane = threadIdx.x;
    const int srow = (lane * 7) & 63;                                    // per-lane source: row (7 lane) mod 64
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + srow * 4),
                                     (__attribute__((address_space(3))) void*)lds, 16, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int e = 0; e < 4; ++e) out[lane * 4 + e] = lds[lane * 4 + e];
}

int main() {
    // ---- 1 ----
    uint16_t* d; (void)hipMalloc(&d, 512);
    tr_probe<<<1, 64>>>(d);
    uint16_t h[256]; (void)hipMemcpy(h, d, 512, hipMemcpyDeviceToHost);
    int ok_tr = 1;
    for (int l = 0; l < 64; ++l) {
        printf("tr16 lane %2d:", l);
        for (int e = 0; e < 4; ++e) {
            printf(" (L%2d,e%d)", h[l * 4 + e] >> 2, h[l * 4 + e] & 3);
            // expectation: within a 16-lane block, lane c element r <- source lane 4 r + c / 4 of the block, element c % 4
            const int blk = l & ~15, c = l & 15;
            if ((h[l * 4 + e] >> 2) != blk + 4 * e + (c >> 2) || (h[l * 4 + e] & 3) != (c & 3)) ok_tr = 0;
        }
        printf("\n");
    }
    printf("TR16_EXPECTED_MAPPING %s\n", ok_tr ? "YES" : "NO");
    // ---- 2 ----
    uint16_t hA[32 * 16], hB[32 * 16];
    for (int i = 0; i < 32; ++i) for (int k = 0; k < 16; ++k) {
        hA[i * 16 + k] = f2bf((float)((i * 7 + k * 3) % 11 - 5));
        hB[i * 16 + k] = f2bf((float)((i * 5 + k * 2 + i * k) % 13 - 6));
    }
    uint16_t *dA, *dB; float* dC;
    (void)hipMalloc(&dA, sizeof(hA)); (void)hipMalloc(&dB, sizeof(hB)); (void)hipMalloc(&dC, 32 * 32 * 4);
    (void)hipMemcpy(dA, hA, sizeof(hA), hipMemcpyHostToDevice); (void)hipMemcpy(dB, hB, sizeof(hB), hipMemcpyHostToDevice);
    mfma_probe<<<1, 64>>>(dA, dB, dC);
    float hC[32 * 32]; (void)hipMemcpy(hC, dC, sizeof(hC), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) {
        float s = 0.f;
        for (int k = 0; k < 16; ++k) s += bf2f(hA[i * 16 + k]) * bf2f(hB[j * 16 + k]);
        if (fabsf(s - hC[i * 32 + j]) > 1e-3f) { if (bad < 5) printf("mfma mismatch C[%d][%d] = %f, want %f\n", i, j, hC[i * 32 + j], s); ++bad; }
    }
    printf("MFMA_32x32x16_LAYOUT %s (%d mismatches)\n", bad ? "NO" : "YES", bad);
    // ---- 3 ----
    uint32_t hs[256], *ds, *dout;
    for (int i = 0; i < 256; ++i) hs[i] = 1000 + i;
    (void)hipMalloc(&ds, 1024); (void)hipMalloc(&dout, 1024);
    (void)hipMemcpy(ds, hs, 1024, hipMemcpyHostToDevice);
    glds_probe<<<1, 64>>>(ds, dout);
    uint32_t ho[256]; (void)hipMemcpy(ho, dout, 1024, hipMemcpyDeviceToHost);
    int bad3 = 0;
    for (int l = 0; l < 64; ++l) for (int e = 0; e < 4; ++e) if (ho[l * 4 + e] != 1000u + ((l * 7) & 63) * 4 + e) ++bad3;
    printf("GLDS_LANE_LINEAR_DEST %s (%d mismatches; lane 1 got %u %u %u %u)\n", bad3 ? "NO" : "YES", bad3, ho[4], ho[5], ho[6], ho[7]);
    printf("hip status: %s\n", hipGetErrorString(hipDeviceSynchronize()));
    return 0;
}
