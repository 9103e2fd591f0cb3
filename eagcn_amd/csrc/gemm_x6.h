// fp32 GEMM tile on the bf16 matrix cores: every fp32 operand element is split EXACTLY into three bf16 pieces
// (a = a0 + a1 + a2: 8 + 8 + 8 significand bits, by truncation of the bit pattern, each remainder computed
// exactly in fp32) and the product is accumulated in fp32 from the six piece products of weight >= 2^-16,
//     a.b  ~=  a0.b0 + a0.b1 + a1.b0 + a0.b2 + a1.b1 + a2.b0          (dropped: a1.b2 + a2.b1 + a2.b2 <= 3 * 2^-24 |a.b|)
// i.e. the same relative error as one fp32 rounding.  v_mfma_f32_16x16x32_bf16 runs at 16x the rate of
// v_mfma_f32_16x16x4_f32 (MI355X_MICROARCH.md), so six of them per 32 k still are 2.7x the fp32 MFMA rate; the
// split (about six VALU operations per element and tile) is done once per tile on the way into LDS.
//
// Same interface as gemm_tile (gemm.hip): one 64 x 64 output tile of split z, operand storage given by
// (A_KC, B_KC), extents from the device.  LDS image per operand and piece: [64 rows][32 k] bf16 (64 bytes per
// row); a lane's MFMA fragment is 8 consecutive k of one row = one ds_read_b128.  The 16-byte k-block q of row
// r is stored at block q ^ ((r >> 1) & 3): conflict-free for the lane groups ds_read_b128 is served in and for
// the 8-lane groups of both ds_write_b128 staging patterns (MI355X_MICROARCH.md, LDS table), without padding.
#pragma once
#include "common.h"

namespace eagcn {

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
constexpr int x6_lds_bytes(int bm, int bn, int np = 3) { return 2 * np * (bm + bn) * 64; }   // two buffers x pieces x rows x 64 B
constexpr int X6_LDS_BYTES = x6_lds_bytes(64, 64);

// eight fp32 values (consecutive k) -> eight bf16 values, round to nearest even (the plain bf16 mode: ONE piece, ONE product;
// operands rounded to 8 significand bits, fp32 accumulation -- BASELINE.json configs[1] as written, NOT the parity path)
__device__ __forceinline__ void bf16_round8(const float (&v)[8], u32x4_t& pl) {
    uint32_t h[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const uint32_t b = __float_as_uint(v[e]);
        h[e] = (b + 0x7FFFu + ((b >> 16) & 1u)) & 0xFFFF0000u;
    }
    pl[0] = (h[0] >> 16) | h[1];
    pl[1] = (h[2] >> 16) | h[3];
    pl[2] = (h[4] >> 16) | h[5];
    pl[3] = (h[6] >> 16) | h[7];
}

// eight fp32 values (consecutive k) -> three vectors of eight bf16 pieces
__device__ __forceinline__ void x6_split8(const float (&v)[8], u32x4_t (&pl)[3]) {
    uint32_t h[3][8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const uint32_t a0 = __float_as_uint(v[e]) & 0xFFFF0000u;
        const float r1 = v[e] - __uint_as_float(a0);                 // exact: at most 16 significant bits
        const uint32_t a1 = __float_as_uint(r1) & 0xFFFF0000u;
        const float r2 = r1 - __uint_as_float(a1);                   // exact: at most 8 significant bits = a bf16
        h[0][e] = a0;
        h[1][e] = a1;
        h[2][e] = __float_as_uint(r2);
    }
#pragma unroll
    for (int p = 0; p < 3; ++p) {
        pl[p][0] = (h[p][0] >> 16) | (h[p][1] & 0xFFFF0000u);
        pl[p][1] = (h[p][2] >> 16) | (h[p][3] & 0xFFFF0000u);
        pl[p][2] = (h[p][4] >> 16) | (h[p][5] & 0xFFFF0000u);
        pl[p][3] = (h[p][6] >> 16) | (h[p][7] & 0xFFFF0000u);
    }
}

// NP = 3: the exact split (six products); NP = 1: plain bf16 operands (one product)
template <bool A_KC, bool B_KC, int BM = 64, int BN = 64, int NP = 3>
__device__ __forceinline__ void gemm_tile_x6(const GemmDesc& g, const int Mx, const int Kx, const int tile_x,
                                             const int tile_y, const int z, const int nsp, unsigned char* smem) {
    constexpr int BK = 32;
    constexpr int WM = BM / 2, WN = BN / 2;                          // 2 x 2 waves
    constexpr int MR = WM / 16, NR = WN / 16;                        // MFMA tiles per wave
    constexpr int APL = BM * 64, BPL = BN * 64;                      // bytes of one piece plane: rows x 32 bf16
    constexpr int BUF = NP * (APL + BPL);
    constexpr int ABLK = BM / 64, BBLK = BN / 64;                    // 8-element blocks a thread stages per operand
    // smem: x6_lds_bytes(BM, BN) of 16-byte aligned LDS owned by the calling kernel: [buffer][A pieces | B pieces]
    auto aplane = [&](int buf, int p) -> unsigned char* { return smem + buf * BUF + p * APL; };
    auto bplane = [&](int buf, int p) -> unsigned char* { return smem + buf * BUF + NP * APL + p * BPL; };

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, q = lane >> 4;
    const int wm = (wave >> 1) * WM, wn = (wave & 1) * WN;
    const int m0 = tile_y * BM, n0 = tile_x * BN;
    const int kchunk = ((Kx + nsp - 1) / nsp + BK - 1) / BK * BK;
    const int kbeg = z * kchunk;
    const int kend = min(Kx, kbeg + kchunk);
    float* C = g.C + (size_t)z * g.slab;

    // loader geometry: a staged block = 8 consecutive k of one row (= one 16-byte LDS block per piece).
    //   K-contiguous storage : block u of a thread: row = 64*u + tid / 4, k-block = tid % 4   (two float4 loads)
    //   MN-contiguous storage: row = 64*u + tid % 64, k-block = tid / 64                      (eight coalesced scalar loads)
    const int a_row = A_KC ? (tid >> 2) : (tid & 63), a_kb = A_KC ? (tid & 3) : (tid >> 6);
    const int b_row = B_KC ? (tid >> 2) : (tid & 63), b_kb = B_KC ? (tid & 3) : (tid >> 6);
    float ra[ABLK][8], rb[BBLK][8];
    auto load_op = [&](const float* __restrict__ P, int ld, bool kc, bool vec, int row, int kb, int r0, int rmax, int k0,
                       float (&r)[8]) {
#pragma unroll
        for (int e = 0; e < 8; ++e) r[e] = 0.0f;
        const int k = k0 + kb * 8;
        if (r0 + row >= rmax || k >= kend) return;
        if (kc) {
            const float* src = P + (size_t)(r0 + row) * ld + k;
            if (vec && k + 8 <= kend) {
                const float4 v0 = *reinterpret_cast<const float4*>(src);
                const float4 v1 = *reinterpret_cast<const float4*>(src + 4);
                r[0] = v0.x; r[1] = v0.y; r[2] = v0.z; r[3] = v0.w;
                r[4] = v1.x; r[5] = v1.y; r[6] = v1.z; r[7] = v1.w;
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e)
                    if (k + e < kend) r[e] = src[e];
            }
        } else {
            const float* src = P + (size_t)k * ld + r0 + row;
#pragma unroll
            for (int e = 0; e < 8; ++e)
                if (k + e < kend) r[e] = src[(size_t)e * ld];
        }
    };
    auto load_tiles = [&](int k0) {
#pragma unroll
        for (int u = 0; u < ABLK; ++u) load_op(g.A, g.lda, A_KC, g.vecA != 0, a_row + 64 * u, a_kb, m0, Mx, k0, ra[u]);
#pragma unroll
        for (int u = 0; u < BBLK; ++u) load_op(g.B, g.ldb, B_KC, g.vecB != 0, b_row + 64 * u, b_kb, n0, g.N, k0, rb[u]);
    };
    // 16-byte block kb of row r lives at block kb ^ ((r >> 1) & 3): conflict-free for the fragment reads and for
    // both staging patterns (tools: the search is in DESIGN.md section 4)
    auto store_tiles = [&](int buf) {
        u32x4_t pl[3];
#pragma unroll
        for (int u = 0; u < ABLK; ++u) {
            if constexpr (NP == 1) bf16_round8(ra[u], pl[0]); else x6_split8(ra[u], pl);
            const int r = a_row + 64 * u;
            const int o = r * 64 + ((a_kb ^ ((r >> 1) & 3)) << 4);
#pragma unroll
            for (int p = 0; p < NP; ++p) *reinterpret_cast<u32x4_t*>(aplane(buf, p) + o) = pl[p];
        }
#pragma unroll
        for (int u = 0; u < BBLK; ++u) {
            if constexpr (NP == 1) bf16_round8(rb[u], pl[0]); else x6_split8(rb[u], pl);
            const int r = b_row + 64 * u;
            const int o = r * 64 + ((b_kb ^ ((r >> 1) & 3)) << 4);
#pragma unroll
            for (int p = 0; p < NP; ++p) *reinterpret_cast<u32x4_t*>(bplane(buf, p) + o) = pl[p];
        }
    };

    f32x4 acc[MR][NR];
#pragma unroll
    for (int i = 0; i < MR; ++i)
#pragma unroll
        for (int j = 0; j < NR; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int nk = (kend > kbeg) ? (kend - kbeg + BK - 1) / BK : 0;
    if (nk > 0) {
        load_tiles(kbeg);
        store_tiles(0);
    }
    __syncthreads();
    // fragment offset inside a row (the swizzle depends on bits 1-2 of the row = bits 1-2 of li)
    const int fo = ((q ^ ((li >> 1) & 3)) << 4);
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nk) load_tiles(kbeg + (kt + 1) * BK);          // lands while this tile is multiplied
        u32x4_t af[MR][NP], bf[NR][NP];
#pragma unroll
        for (int p = 0; p < NP; ++p) {
#pragma unroll
            for (int i = 0; i < MR; ++i) af[i][p] = *reinterpret_cast<const u32x4_t*>(aplane(cur, p) + (wm + i * 16 + li) * 64 + fo);
#pragma unroll
            for (int j = 0; j < NR; ++j) bf[j][p] = *reinterpret_cast<const u32x4_t*>(bplane(cur, p) + (wn + j * 16 + li) * 64 + fo);
        }
        // six piece products per output tile, smallest first; consecutive MFMAs go to DIFFERENT accumulators
#define EAGCN_X6_PROD(PA, PB)                                                                                         \
    _Pragma("unroll") for (int i = 0; i < MR; ++i) _Pragma("unroll") for (int j = 0; j < NR; ++j)                   \
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, af[i][PA]),                 \
                                                            __builtin_bit_cast(bf16x8_t, bf[j][PB]), acc[i][j], 0, 0, 0);
        if constexpr (NP == 3) {
            EAGCN_X6_PROD(2, 0)
            EAGCN_X6_PROD(1, 1)
            EAGCN_X6_PROD(0, 2)
            EAGCN_X6_PROD(1, 0)
            EAGCN_X6_PROD(0, 1)
        }
        EAGCN_X6_PROD(0, 0)
#undef EAGCN_X6_PROD
        if (kt + 1 < nk) store_tiles(cur ^ 1);
        __syncthreads();
    }
    // epilogue: D layout col = lane&15, row = (lane>>4)*4 + reg
#pragma unroll
    for (int i = 0; i < MR; ++i)
#pragma unroll
        for (int j = 0; j < NR; ++j) {
            const int col = n0 + wn + j * 16 + li;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = m0 + wm + i * 16 + q * 4 + r;
                if (row < Mx && col < g.N) C[(size_t)row * g.ldc + col] = acc[i][j][r];
            }
        }
}

}  // namespace eagcn
