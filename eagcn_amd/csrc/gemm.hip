// fp32 GEMM on the CDNA4 matrix cores: v_mfma_f32_16x16x4_f32 (f32 in, f32 accumulate, exact f32;
// bitwise a k-ordered fmaf chain), LDS-staged, register-prefetched, double-buffered.
//
// Used for the flat feature transform of the hot path -- reference layers.py:40 (`torch.mm` over
// all B*N rows) restated over the packed rows only -- and for its two backward products
// (dX = dP.W^T, dW = X^T.dP, the latter split-K over the rows).  Kernels in this file:
//   gemm_f32_kernel       one product; XCD-aware tile order; row / reduction extents from device memory
//   gemm_f32_pair_kernel  dX and dW of a layer in ONE grid (fills the partly empty last round of each)
// Each takes its 64x64 tile from gemm_tile (fp32 MFMA, exact fp32 products) or, opt-in, gemm_tile_x6
// (gemm_x6.h: exact three-way bf16 split, six bf16 MFMA products, fp32 accumulation).
//
// Operand storage is described by (ta, tb): ta = 0 -> A is [M][K] (K contiguous), ta = 1 -> A is
// stored [K][M]; tb = 0 -> B is [K][N] (N contiguous), tb = 1 -> B is stored [N][K].
// LDS images: a K-contiguous operand is kept [row][BK+2] (stride = 2*odd: the 16 rows x 2 k of a
// 32-lane ds_read_b32 group hit 32 distinct banks -- rows land on the 16 even banks, k+1 on the odd
// ones; BK+1 measured 33-50 % bank-conflict cycles); an MN-contiguous operand is kept [k][R+16]
// (stride = 16 mod 32: lanes 0-15 and 16-31 of a group land on disjoint bank halves).
#include <stdlib.h>

#include <algorithm>

#include "common.h"
#include "gemm_x6.h"

namespace eagcn {

// floats of LDS one tile needs (two buffers of an A and a B image)
constexpr int gemm_lds_floats(int BM, int BN, int BK, bool a_kc, bool b_kc) {
    return 2 * ((a_kc ? BM * (BK + 2) : BK * (BM + 16)) + (b_kc ? BN * (BK + 2) : BK * (BN + 16)));
}

// one BM x BN output tile of split z; Mx / Kx are the actual extents (<= g.M / g.K)
template <int BM, int BN, int BK, bool A_KC, bool B_KC, int D, bool EXT_LDS = false>
__device__ __forceinline__ void gemm_tile(const GemmDesc& g, const int Mx, const int Kx, const int tile_x,
                                          const int tile_y, const int z, const int nsp, float* shared = nullptr) {
    constexpr int WM = BM / 2, WN = BN / 2;      // 2x2 waves
    constexpr int MR = WM / 16, NR = WN / 16;
    constexpr int LDA_S = A_KC ? (BK + 2) : (BM + 16);
    constexpr int LDB_S = B_KC ? (BK + 2) : (BN + 16);
    constexpr int A_SZ = A_KC ? BM * LDA_S : BK * LDA_S;
    constexpr int B_SZ = B_KC ? BN * LDB_S : BK * LDB_S;
    // loader geometry
    constexpr int A_TPR = A_KC ? BK / 4 : BM / 4;     // threads per contiguous run
    constexpr int A_RPP = 256 / A_TPR;                 // runs per pass
    constexpr int A_PASS = (A_KC ? BM : BK) / A_RPP;
    constexpr int B_TPR = B_KC ? BK / 4 : BN / 4;
    constexpr int B_RPP = 256 / B_TPR;
    constexpr int B_PASS = (B_KC ? BN : BK) / B_RPP;
    static_assert(A_PASS >= 1 && B_PASS >= 1, "tile too small for 256 threads");

    // `shared`: gemm_lds_floats(...) floats of 16-byte aligned LDS owned by the calling kernel (kernels that hold
    // two differently laid out tiles share ONE buffer that way); by default the tile owns its own
    float* smem;
    if constexpr (EXT_LDS) {
        smem = shared;
    } else {
        __shared__ __attribute__((aligned(16))) float own_smem[2 * (A_SZ + B_SZ)];
        smem = own_smem;
    }
    // K-contiguous images, BK = 16: the 16-lane groups a ds_write_b64 is served in (4 rows x 4 k-quads) take rows
    // {r, r+1, r+8, r+9} instead of {r .. r+3}: with the 18-word row stride rows r and r+2 share banks, rows r and
    // r+8 do not (SQ_LDS_BANK_CONFLICT was 14 % of the LDS cycles of these kernels, all of it from these stores)
    auto kc_row = [](int x) -> int {
        if constexpr (BK == 16) {
            const int g = x >> 2, rs = x & 3;
            return ((g >> 2) << 4) + ((g & 3) << 1) + (rs & 1) + ((rs >> 1) << 3);
        } else {
            return x;
        }
    };
    auto As = [&](int buf) -> float* { return smem + buf * (A_SZ + B_SZ); };
    auto Bs = [&](int buf) -> float* { return smem + buf * (A_SZ + B_SZ) + A_SZ; };

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, q = lane >> 4;
    const int wm = (wave >> 1) * WM, wn = (wave & 1) * WN;
    const int m0 = tile_y * BM, n0 = tile_x * BN;
    // split-K range
    const int kchunk = ((Kx + nsp - 1) / nsp + BK - 1) / BK * BK;
    const int kbeg = z * kchunk;
    const int kend = min(Kx, kbeg + kchunk);
    float* C = g.C + (size_t)z * g.slab;

    // D register stages: tile kt+D is requested while tile kt is being multiplied, so D-1 global->LDS
    // round trips overlap each k-tile (with one stage a short-K product is a chain of exposed round
    // trips: 0.68 us per k-tile measured on 256x256x700)
    float4 ra_[D][A_PASS], rb_[D][B_PASS];
    auto load_tiles = [&](int k0, float4 (&ra)[A_PASS], float4 (&rb)[B_PASS]) {
#pragma unroll
        for (int p = 0; p < A_PASS; ++p) {
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if constexpr (A_KC) {
                int row = p * A_RPP + kc_row(tid / A_TPR), kq = (tid % A_TPR) * 4;
                if (m0 + row < Mx && k0 + kq < kend) {
                    const float* src = g.A + (size_t)(m0 + row) * g.lda + k0 + kq;
                    if (g.vecA) v = *reinterpret_cast<const float4*>(src);
                    else {
                        v.x = src[0];
                        if (k0 + kq + 1 < kend) v.y = src[1];
                        if (k0 + kq + 2 < kend) v.z = src[2];
                        if (k0 + kq + 3 < kend) v.w = src[3];
                    }
                }
            } else {
                int k = p * A_RPP + tid / A_TPR, mq = (tid % A_TPR) * 4;
                if (k0 + k < kend && m0 + mq < Mx) {
                    const float* src = g.A + (size_t)(k0 + k) * g.lda + m0 + mq;
                    if (g.vecA) v = *reinterpret_cast<const float4*>(src);
                    else {
                        v.x = src[0];
                        if (m0 + mq + 1 < Mx) v.y = src[1];
                        if (m0 + mq + 2 < Mx) v.z = src[2];
                        if (m0 + mq + 3 < Mx) v.w = src[3];
                    }
                }
            }
            ra[p] = v;
        }
#pragma unroll
        for (int p = 0; p < B_PASS; ++p) {
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if constexpr (B_KC) {
                int row = p * B_RPP + kc_row(tid / B_TPR), kq = (tid % B_TPR) * 4;
                if (n0 + row < g.N && k0 + kq < kend) {
                    const float* src = g.B + (size_t)(n0 + row) * g.ldb + k0 + kq;
                    if (g.vecB) v = *reinterpret_cast<const float4*>(src);
                    else {
                        v.x = src[0];
                        if (k0 + kq + 1 < kend) v.y = src[1];
                        if (k0 + kq + 2 < kend) v.z = src[2];
                        if (k0 + kq + 3 < kend) v.w = src[3];
                    }
                }
            } else {
                int k = p * B_RPP + tid / B_TPR, nq = (tid % B_TPR) * 4;
                if (k0 + k < kend && n0 + nq < g.N) {
                    const float* src = g.B + (size_t)(k0 + k) * g.ldb + n0 + nq;
                    if (g.vecB) v = *reinterpret_cast<const float4*>(src);
                    else {
                        v.x = src[0];
                        if (n0 + nq + 1 < g.N) v.y = src[1];
                        if (n0 + nq + 2 < g.N) v.z = src[2];
                        if (n0 + nq + 3 < g.N) v.w = src[3];
                    }
                }
            }
            rb[p] = v;
        }
    };
    auto store_tiles = [&](int buf, const float4 (&ra)[A_PASS], const float4 (&rb)[B_PASS]) {
#pragma unroll
        for (int p = 0; p < A_PASS; ++p) {
            if constexpr (A_KC) {
                int row = p * A_RPP + kc_row(tid / A_TPR), kq = (tid % A_TPR) * 4;
                float2* d = reinterpret_cast<float2*>(As(buf) + row * LDA_S + kq);   // 8-byte aligned: LDA_S even
                d[0] = make_float2(ra[p].x, ra[p].y);
                d[1] = make_float2(ra[p].z, ra[p].w);
            } else {
                int k = p * A_RPP + tid / A_TPR, mq = (tid % A_TPR) * 4;
                *reinterpret_cast<float4*>(As(buf) + k * LDA_S + mq) = ra[p];
            }
        }
#pragma unroll
        for (int p = 0; p < B_PASS; ++p) {
            if constexpr (B_KC) {
                int row = p * B_RPP + kc_row(tid / B_TPR), kq = (tid % B_TPR) * 4;
                float2* d = reinterpret_cast<float2*>(Bs(buf) + row * LDB_S + kq);
                d[0] = make_float2(rb[p].x, rb[p].y);
                d[1] = make_float2(rb[p].z, rb[p].w);
            } else {
                int k = p * B_RPP + tid / B_TPR, nq = (tid % B_TPR) * 4;
                *reinterpret_cast<float4*>(Bs(buf) + k * LDB_S + nq) = rb[p];
            }
        }
    };

    f32x4 acc[MR][NR];
#pragma unroll
    for (int i = 0; i < MR; ++i)
#pragma unroll
        for (int j = 0; j < NR; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int nk = (kend > kbeg) ? (kend - kbeg + BK - 1) / BK : 0;
#pragma unroll
    for (int d = 0; d < D; ++d)
        if (d < nk) load_tiles(kbeg + d * BK, ra_[d], rb_[d]);
    if (nk > 0) store_tiles(0, ra_[0], rb_[0]);
    __syncthreads();
    for (int kt0 = 0; kt0 < nk; kt0 += D) {
#pragma unroll
        for (int d = 0; d < D; ++d) {             // unrolled so that every stage index is a compile-time constant
            const int kt = kt0 + d;
            if (kt < nk) {
                const int cur = kt & 1;
                // stage d held tile kt, which is already in LDS: reuse it for tile kt+D
                if (kt + D < nk) load_tiles(kbeg + (kt + D) * BK, ra_[d], rb_[d]);
                const float* a_s = As(cur);
                const float* b_s = Bs(cur);
#pragma unroll
                for (int ks = 0; ks < BK / 4; ++ks) {
                    float af[MR], bf[NR];
#pragma unroll
                    for (int i = 0; i < MR; ++i) {
                        if constexpr (A_KC) af[i] = a_s[(wm + i * 16 + li) * LDA_S + ks * 4 + q];
                        else af[i] = a_s[(ks * 4 + q) * LDA_S + wm + i * 16 + li];
                    }
#pragma unroll
                    for (int j = 0; j < NR; ++j) {
                        if constexpr (B_KC) bf[j] = b_s[(wn + j * 16 + li) * LDB_S + ks * 4 + q];
                        else bf[j] = b_s[(ks * 4 + q) * LDB_S + wn + j * 16 + li];
                    }
#pragma unroll
                    for (int i = 0; i < MR; ++i)
#pragma unroll
                        for (int j = 0; j < NR; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i], bf[j], acc[i][j], 0, 0, 0);
                }
                if (kt + 1 < nk) store_tiles(cur ^ 1, ra_[(d + 1) % D], rb_[(d + 1) % D]);
                __syncthreads();
            }
        }
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

// Split-K factor actually used.  When the K extent is only known on the device (dW = X^T.dP over the packed
// rows, sized on the host for the row CAPACITY) every split keeps at least 8 k-tiles of 16: the remaining
// partial slabs are neither written nor read (unpack_grads applies the same rule, eagcn_eff_splits).
__device__ __forceinline__ int eff_splits(const GemmDesc& g, int Kx) {
    return g.K_dev ? max(1, min(g.splits, Kx >> 7)) : g.splits;
}

// XCD-aware order of `nwg` real workgroups: dispatch slot `lin` runs on XCD lin % 8; give every XCD a contiguous
// run of tiles.  Bijective on [0, nwg).
__device__ __forceinline__ int xcd_remap(int lin, int nwg) {
    const int xcd = lin & 7, qn = nwg >> 3, rn = nwg & 7;
    return (xcd < rn ? xcd * (qn + 1) : rn * (qn + 1) + (xcd - rn) * qn) + (lin >> 3);
}

template <int BM, int BN, int BK, bool A_KC, bool B_KC, int D, int X6 = 0>
__global__ __launch_bounds__(256) void gemm_f32_kernel(GemmDesc g) {
    // extents that are known only on the device (packed row count); locals, never written back into
    // the by-value argument block (that would demote it to scratch memory)
    const int Mx = g.M_dev ? min(*g.M_dev, g.M) : g.M;
    const int Kx = g.K_dev ? min(*g.K_dev, g.K) : g.K;
    // XCD-aware tile order (workgroup b runs on XCD b % 8, each XCD has a private L2): every XCD gets a
    // contiguous run of the REAL tiles (the grid may be sized for a row capacity), so the column tiles
    // that share an A row panel hit the same L2.  Bijective for any tile count; affects speed only.
    const int nsp = eff_splits(g, Kx);
    int tile_x, tile_y, z;
    {
        const int gx = gridDim.x;
        const int per_z = gx * ((Mx + BM - 1) / BM);           // tiles that have rows, per split
        const int nwg = per_z * nsp;
        if (blockIdx.y * gx + blockIdx.x >= per_z || (int)blockIdx.z >= nsp) return;   // uniform: capacity-only workgroup
        // linear dispatch index of the REAL workgroups (capacity-only ones were skipped above)
        const int lin = blockIdx.z * per_z + blockIdx.y * gx + blockIdx.x;
        const int nid = xcd_remap(lin, nwg);
        z = nid / per_z;                                       // all tiles of one K-split land on one or two XCDs
        const int rem = nid - z * per_z;
        tile_y = rem / gx;
        tile_x = rem - tile_y * gx;
    }
    if constexpr (X6 != 0) {
        constexpr int NP = X6 == 2 ? 1 : 3;
        __shared__ __attribute__((aligned(16))) unsigned char x6_smem[x6_lds_bytes(BM, BN, NP)];
        gemm_tile_x6<A_KC, B_KC, BM, BN, NP>(g, Mx, Kx, tile_x, tile_y, z, nsp, x6_smem);
    }
    else gemm_tile<BM, BN, BK, A_KC, B_KC, D>(g, Mx, Kx, tile_x, tile_y, z, nsp);
}

// The two backward products of a layer, dX = dP.W^T (rows = packed rows, capacity-sized) and dW = X^T.dP
// (split-K over the rows), in ONE grid: each alone leaves a partly filled last round of workgroups (532 tiles on
// 256 CUs); together the second fills the tail of the first.  Workgroups [0, first1) belong to dX.
template <int BM, int BN, int BK, int D, int X6 = 0>
__global__ __launch_bounds__(256) void gemm_f32_pair_kernel(GemmDesc g0, GemmDesc g1, int first1) {
    // one LDS buffer for both halves of the grid
    constexpr int F32_LDS = gemm_lds_floats(BM, BN, BK, true, true) > gemm_lds_floats(BM, BN, BK, false, false)
                                ? gemm_lds_floats(BM, BN, BK, true, true) : gemm_lds_floats(BM, BN, BK, false, false);
    constexpr int NP = X6 == 2 ? 1 : 3;
    __shared__ __attribute__((aligned(16))) unsigned char x6_smem[X6 ? x6_lds_bytes(64, 64, NP) : F32_LDS * 4];
    float* f32_smem = reinterpret_cast<float*>(x6_smem);
    const int b = blockIdx.x;
    if (b < first1) {
        const int Mx = g0.M_dev ? min(*g0.M_dev, g0.M) : g0.M;
        const int gx = (g0.N + BN - 1) / BN;
        const int nreal = gx * ((Mx + BM - 1) / BM);
        if (b >= nreal) return;
        const int nid = xcd_remap(b, nreal);
        const int ty = nid / gx;
        if constexpr (X6 != 0) gemm_tile_x6<true, true, 64, 64, NP>(g0, Mx, g0.K, nid - ty * gx, ty, 0, 1, x6_smem);
        else gemm_tile<BM, BN, BK, true, true, D, true>(g0, Mx, g0.K, nid - ty * gx, ty, 0, 1, f32_smem);
    } else {
        const int lin = b - first1;                                 // first1 is a multiple of 8
        const int Kx = g1.K_dev ? min(*g1.K_dev, g1.K) : g1.K;
        const int gx = (g1.N + BN - 1) / BN;
        const int per_z = gx * ((g1.M + BM - 1) / BM);
        const int nsp = eff_splits(g1, Kx);
        const int nwg = per_z * nsp;
        if (lin >= nwg) return;
        const int nid = xcd_remap(lin, nwg);
        const int z = nid / per_z, rem = nid - z * per_z;
        const int ty = rem / gx;
        if constexpr (X6 != 0) gemm_tile_x6<false, false, 64, 64, NP>(g1, g1.M, Kx, rem - ty * gx, ty, z, nsp, x6_smem);
        else gemm_tile<BM, BN, BK, false, false, D, true>(g1, g1.M, Kx, rem - ty * gx, ty, z, nsp, f32_smem);
    }
}


template <int BM, int BN, int BK, int D>
static int launch_cfg(const GemmDesc& g, hipStream_t s) {
    dim3 grid(cdiv(g.N, BN), cdiv(g.M, BM), g.splits);
    ProfScope ps(g.prof_tag, s, g.work > 0.0 ? g.work : 2.0 * g.M * g.N * g.K);
    if (g.ta == 0 && g.tb == 0) gemm_f32_kernel<BM, BN, BK, true, false, D><<<grid, 256, 0, s>>>(g);
    else if (g.ta == 0 && g.tb == 1) gemm_f32_kernel<BM, BN, BK, true, true, D><<<grid, 256, 0, s>>>(g);
    else if (g.ta == 1 && g.tb == 0) gemm_f32_kernel<BM, BN, BK, false, false, D><<<grid, 256, 0, s>>>(g);
    else gemm_f32_kernel<BM, BN, BK, false, true, D><<<grid, 256, 0, s>>>(g);
    EAGCN_LAUNCH_CHECK();
    return EAGCN_OK;
}

// Which matrix-core path the layer products take: 0 (default) = fp32 MFMA, exact fp32 products; 1 = bf16 x 6
// (gemm_x6.h) for products with K >= 64.  Set by EAGCN_GEMM_X6 at load time or eagcn_set_gemm_mode() at run time.
// Measured on MI355X (tools/gemm_bench.py, gemm_accuracy.py): same accuracy (4-6 eps of sum|a||b|, as
// torch.matmul), dX 41 vs 55 us and forward 41 vs 44 us at the Tox21 shape, 92 vs 78 TF at 20 k rows -- a 2.5 %
// step-time gain, so the exact-product path stays the default.
// 2 = plain bf16 operands (rounded to nearest even on the way into LDS, ONE bf16 MFMA product, fp32 accumulation): the
// "bf16" of BASELINE.json configs[1]; NOT the parity path (products carry 2^-9 relative rounding per operand).
// default 3: operand planes written by the producers, products on the bf16 matrix cores (gemm_bx3.hip) -- the whole GPU parity
// suite is green in this mode with the tolerances of the fp32 MFMA path (profiles/r04_parity_report.txt)
static int g_gemm_x6 = [] { const char* e = getenv("EAGCN_GEMM_X6"); return e ? atoi(e) : 3; }();
// (modes 3 / 4 -- operand planes written by the producers, gemm_bx3.hip -- are decided per layer in layer.hip; whatever still
//  reaches this file in those modes runs on the fp32 matrix path)
static int use_x6(const GemmDesc& g) { return ((g_gemm_x6 == 1 || g_gemm_x6 == 2) && g.K >= 64 && g.prof_tag != PROF_HEAD) ? g_gemm_x6 : 0; }
int gemm_mode() { return g_gemm_x6; }
template <int BM, int BN, int X6>
static int launch_x6_cfg(const GemmDesc& g, hipStream_t s) {
    dim3 grid(cdiv(g.N, BN), cdiv(g.M, BM), g.splits);
    ProfScope ps(g.prof_tag, s, g.work > 0.0 ? g.work : 2.0 * g.M * g.N * g.K);
    if (g.ta == 0 && g.tb == 0) gemm_f32_kernel<BM, BN, 16, true, false, 4, X6><<<grid, 256, 0, s>>>(g);
    else if (g.ta == 0 && g.tb == 1) gemm_f32_kernel<BM, BN, 16, true, true, 4, X6><<<grid, 256, 0, s>>>(g);
    else if (g.ta == 1 && g.tb == 0) gemm_f32_kernel<BM, BN, 16, false, false, 4, X6><<<grid, 256, 0, s>>>(g);
    else gemm_f32_kernel<BM, BN, 16, false, true, 4, X6><<<grid, 256, 0, s>>>(g);
    EAGCN_LAUNCH_CHECK();
    return EAGCN_OK;
}
static int launch_x6(const GemmDesc& g, hipStream_t s, int mode) {
    if (mode == 2) return launch_x6_cfg<64, 64, 2>(g, s);
    // EAGCN_X6_TILE = 0: 64x64, 1: 128x64, 2: 128x128 (tools/gemm_bench.py)
    static const int tile = [] { const char* e = getenv("EAGCN_X6_TILE"); return e ? atoi(e) : 0; }();
    if (tile == 2 && g.M > 64 && g.N > 64) return launch_x6_cfg<128, 128, 1>(g, s);
    if (tile >= 1 && g.M > 64) return launch_x6_cfg<128, 64, 1>(g, s);
    return launch_x6_cfg<64, 64, 1>(g, s);
}

int launch_gemm(const GemmDesc& g0, hipStream_t s) {
    if (g0.M <= 0 || g0.N <= 0) return EAGCN_OK;
    EAGCN_CHECK_ARG(g0.splits >= 1, "gemm: splits must be >= 1");
    GemmDesc g = g0;
    // contiguous runs are loaded as float4 when every run is 16-byte aligned and whole, else scalar
    g.vecA = (g.lda % 4) == 0 && ((g.ta ? g.M : g.K) % 4) == 0 && (reinterpret_cast<uintptr_t>(g.A) % 16) == 0;
    g.vecB = (g.ldb % 4) == 0 && ((g.tb ? g.K : g.N) % 4) == 0 && (reinterpret_cast<uintptr_t>(g.B) % 16) == 0;
    if (g.splits > 1) EAGCN_CHECK_ARG(g.vecA && g.vecB, "gemm: split-K needs float4-aligned operands");
    // tile choice: EAGCN_GEMM_CFG=<id> forces one configuration (tools/gemm_bench.py)
    static const int forced = [] { const char* e = getenv("EAGCN_GEMM_CFG"); return e ? atoi(e) : -1; }();
    int cfg = forced;
    if (cfg < 0 && use_x6(g)) return launch_x6(g, s, use_x6(g));
    if (cfg < 0) {
        const long tiles128 = (long)cdiv(g.M, 128) * cdiv(g.N, 128) * g.splits;
        // measured on MI355X (tools/gemm_bench.py, gemm_big.py): 128x128 tiles win only for long-K, many-tile
        // problems (4096^3: 113 vs 98 TF); at the layer shapes (K = 24..704) 64x64 is equal or better, and in
        // graph mode M is only a capacity, so do not let it pick the large tile
        cfg = (g.K >= 1024 && tiles128 >= 512 && g.M > 64 && g.N > 64) ? 3 : 0;
    }
    switch (cfg) {
        case 1: return launch_cfg<64, 64, 32, 3>(g, s);
        case 2: return launch_cfg<64, 64, 16, 2>(g, s);
        case 3: return launch_cfg<128, 128, 16, 3>(g, s);
        case 4: return launch_cfg<64, 64, 16, 6>(g, s);
        case 5: return launch_cfg<64, 64, 16, 1>(g, s);
        default: return launch_cfg<64, 64, 16, 4>(g, s);
    }
}

int launch_gemm_pair(const GemmDesc& dx, const GemmDesc& dw, hipStream_t s) {
    GemmDesc g0 = dx, g1 = dw;
    g0.vecA = (g0.lda % 4) == 0 && (g0.K % 4) == 0 && (reinterpret_cast<uintptr_t>(g0.A) % 16) == 0;
    g0.vecB = (g0.ldb % 4) == 0 && (g0.K % 4) == 0 && (reinterpret_cast<uintptr_t>(g0.B) % 16) == 0;
    g1.vecA = (g1.lda % 4) == 0 && (g1.M % 4) == 0 && (reinterpret_cast<uintptr_t>(g1.A) % 16) == 0;
    g1.vecB = (g1.ldb % 4) == 0 && (g1.N % 4) == 0 && (reinterpret_cast<uintptr_t>(g1.B) % 16) == 0;
    const bool ok = g0.ta == 0 && g0.tb == 1 && g0.splits == 1 && !g0.K_dev && g1.ta == 1 && g1.tb == 0 && !g1.M_dev &&
                    (g1.splits == 1 || (g1.vecA && g1.vecB)) && g0.M > 0 && g0.N > 0 && g1.M > 0 && g1.N > 0;
    if (!ok) {
        int rc = launch_gemm(dw, s);
        return rc ? rc : launch_gemm(dx, s);
    }
    const int first1 = (cdiv(g0.M, 64) * cdiv(g0.N, 64) + 7) / 8 * 8;
    const int n1 = cdiv(g1.M, 64) * cdiv(g1.N, 64) * g1.splits;
    const double w0 = g0.work > 0.0 ? g0.work : 2.0 * g0.M * g0.N * g0.K;
    const double w1 = g1.work > 0.0 ? g1.work : 2.0 * g1.M * g1.N * g1.K;
    ProfScope ps(PROF_GEMM_PAIR, s, w0 + w1);
    if (use_x6(g0) == 2 && use_x6(g1) == 2) gemm_f32_pair_kernel<64, 64, 16, 4, 2><<<first1 + n1, 256, 0, s>>>(g0, g1, first1);
    else if (use_x6(g0) && use_x6(g1)) gemm_f32_pair_kernel<64, 64, 16, 4, 1><<<first1 + n1, 256, 0, s>>>(g0, g1, first1);
    else gemm_f32_pair_kernel<64, 64, 16, 4><<<first1 + n1, 256, 0, s>>>(g0, g1, first1);
    EAGCN_LAUNCH_CHECK();
    return EAGCN_OK;
}

}  // namespace eagcn

using namespace eagcn;

extern "C" int eagcn_set_gemm_mode(int mode) {
    const int old = eagcn::g_gemm_x6;
    eagcn::g_gemm_x6 = mode;
    return old;
}

extern "C" int eagcn_gemm_f32(int ta, int tb, int M, int N, int K, const float* A, int lda, const float* B,
                              int ldb, float* C, int ldc, void* stream) {
    EAGCN_CHECK_ARG(A && B && C, "eagcn_gemm_f32: null operand");
    GemmDesc g{ta, tb, M, N, K, A, lda, B, ldb, C, ldc, 1, 0};
    return launch_gemm(g, (hipStream_t)stream);
}

// The same product with ONE extent taken from device memory (M and K are then capacities: grid and strides): which = 0: the
// rows of A / C, which = 2: the reduction length.  Products over the packed rows of a capacity-sized batch index (graph mode).
extern "C" int eagcn_gemm_f32_dev(int ta, int tb, int M, int N, int K, const float* A, int lda, const float* B,
                                  int ldb, float* C, int ldc, const int32_t* extent_dev, int which, void* stream) {
    EAGCN_CHECK_ARG(A && B && C && extent_dev && (which == 0 || which == 2), "eagcn_gemm_f32_dev: bad argument");
    GemmDesc g{ta, tb, M, N, K, A, lda, B, ldb, C, ldc, 1, 0};
    if (which == 0) g.M_dev = extent_dev; else g.K_dev = extent_dev;
    return launch_gemm(g, (hipStream_t)stream);
}
