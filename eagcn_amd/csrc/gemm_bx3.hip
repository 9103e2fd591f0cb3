// Exact-fp32 layer products at the bf16 matrix rate: operands arrive as THREE bf16 PLANES written by their producers.
//
// An fp32 value splits exactly into three bf16 pieces, x = x0 + x1 + x2 (8 + 8 + 8 significand bits, by truncation; bx3_split
// below), and a product is rebuilt from the six piece products of weight >= 2^-16 with fp32 accumulation,
//     a.b ~= a2.b0 + a1.b1 + a0.b2 + a1.b0 + a0.b1 + a0.b0          (dropped: a1.b2 + a2.b1 + a2.b2 <= 3 * 2^-24 |a.b|),
// i.e. one fp32 rounding worth of error per product -- what csrc/gemm_x6.h established in round 2 (tools/gemm_accuracy.py: 4-6
// eps of sum |a||b|, the same as the fp32 MFMA path and torch.matmul).  That kernel lost to the fp32 MFMA kernel (gemm3.hip)
// because every CONSUMER re-split its fp32 operands on the way into LDS (six VALU operations per element and tile, three LDS
// planes written through ds_write).  Here the split happens ONCE, in the kernel that produces the matrix (bn_apply for X,
// agg_edge for dP, pack_params for the weights): the GEMM never sees an fp32 operand, never executes a VALU instruction per
// element, and moves its tiles global -> LDS by LDS-DMA (buffer_load ... lds: no staging registers, no ds_write).
// v_mfma_f32_32x32x16_bf16 runs at 16x the rate of v_mfma_f32_16x16x4_f32 (MI355X_MICROARCH.md), so six of them per 16 k are
// 2.7x the fp32 matrix rate: the ceiling of this file is 2.5 PF / 6 = 417 TFLOP/s of fp32-equivalent work.
//
// Forms (reference layers.py:40 `torch.mm(x, W)` and its two autograd products):
//   NT  C[M,N] = A[M,K] . B[N,K]^T   both operands K-contiguous: forward (X . WcatT^T) and dX = dP . Wcat^T
//   TN  C[M,N] = A[K,M]^T . B[K,N]   both operands K-major:      dW = X^T . dP, K = packed rows, split-K over `splits` slabs
// Workgroup tile (64 WM) x (64 WN) x 32, one wave per 64 x 64 (2 x 2 MFMA tiles of 32 x 32, 64 accumulator registers).
// LDS image of one operand plane and k-tile:
//   NT  [rows][32 k]  = 64 bytes per row; 16-byte chunk c of row r sits at chunk c ^ ((r >> 2) & 3)
//   TN  [32 k][rows]  = 2 rows bytes per k;  16-byte chunk c of k-row r sits at chunk c ^ ((r & 3) << 2)
// both conflict-free for the lane groups their fragment reads are served in (ds_read_b128 / ds_read_b64_tr_b16: the transposing
// read hands a lane the four k-consecutive values of ITS column out of a [4 k][16 column] block -- tools/probes/bx3_probe.hip
// pins the mapping).  LDS-DMA writes lane-linear (base + 16 lane), so the swizzle lives in the per-lane SOURCE address.
// Out-of-range rows (M / K tails, device-side extents) read as ZERO through the range check of the buffer descriptors, whose
// num_records are built from the actual extents.
// Pipeline: NS LDS stages (3 for 128 x 128, 2 for the 8-wave tiles); tile kt + NS - 1 is in flight while tile kt is multiplied;
// one raw s_barrier and one counted s_waitcnt vmcnt per k-tile, nothing else.
// Scheduling: PERSISTENT -- one workgroup per CU walks a contiguous run of work units (output tiles, or (tile, k-chunk) items of
// a split-K product) counted on the device from the actual extents; every XCD (dispatch slot b runs on XCD b % 8) owns a
// contiguous eighth of the unit list, so the row panels of neighbouring tiles meet in ONE L2.
#include <stdlib.h>

#include <algorithm>

#include "bx3.h"
#include "common.h"
#include "kernels.h"

namespace eagcn {

typedef __bf16 bx_bf16x8 __attribute__((ext_vector_type(8)));
typedef float bx_f32x16 __attribute__((ext_vector_type(16)));
typedef uint32_t bx_u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t bx_u32x2 __attribute__((ext_vector_type(2)));
typedef short bx_s16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* bx_lds_ptr;

constexpr int BX_BK = 32;

extern __shared__ __attribute__((aligned(1024))) unsigned char bx_smem[];

template <int WM, int WN>
struct BxCfg {
    static constexpr int NW = WM * WN;
    static constexpr int BM = 64 * WM, BN = 64 * WN;
    static constexpr int APL = BM * 64, BPL = BN * 64;             // bytes of one plane image of a k-tile
    static constexpr int STAGE = 3 * (APL + BPL);
    static constexpr int NS = (3 * STAGE <= 160 * 1024) ? 3 : 2;
    static constexpr int APW = BM / 16 / NW, BPW = BN / 16 / NW;   // 1 KB DMA pieces per wave and plane
    static constexpr int PPT = 3 * (APW + BPW);                    // DMA instructions per wave and k-tile
    static_assert(APW >= 1 && BPW >= 1 && (BM / 16) % NW == 0 && (BN / 16) % NW == 0, "tile too small for the wave count");
};

template <int N>
__device__ __forceinline__ void bx_wait_vm() {
    static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit counter");
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// One output tile (or one k-chunk of it): k-tiles [kt0, kt1) of BX_BK.  NP = 3: exact split; NP = 1: plain bf16 operands.
template <bool TN, int WM, int WN, int NP, int DBG = 0>
__device__ __forceinline__ void bx_tile(const BxProb& p, const int Mx, const int Kx, const int tm, const int tn, const int kt0,
                                        const int kt1, float* __restrict__ Cz) {
    using Cf = BxCfg<WM, WN>;
    constexpr int BM = Cf::BM, BN = Cf::BN, NS = Cf::NS, APW = Cf::APW, BPW = Cf::BPW;
    constexpr int PPT = NP * (APW + BPW);
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wave / WN, wn = wave - wm * WN;
    const int m0 = tm * BM, n0 = tn * BN;

    // ---- buffer descriptors: one per operand plane, num_records from the ACTUAL extents (rows beyond them read as zero) --------
    // NT: A rows = M, B rows = N;  TN: rows of both = K
    const unsigned arows = TN ? (unsigned)Kx : (unsigned)Mx;
    const unsigned brows = TN ? (unsigned)Kx : (unsigned)p.N;
    __amdgpu_buffer_rsrc_t ra[NP], rb[NP];
#pragma unroll
    for (int q = 0; q < NP; ++q) {
        ra[q] = __builtin_amdgcn_make_buffer_rsrc((void*)(p.A.p + (size_t)q * p.A.pstride), 0, (int)(arows * (unsigned)p.A.ld * 2u), 0x00020000);
        rb[q] = __builtin_amdgcn_make_buffer_rsrc((void*)(p.B.p + (size_t)q * p.B.pstride), 0, (int)(brows * (unsigned)p.B.ld * 2u), 0x00020000);
    }

    // ---- DMA source offsets (bytes) of this lane's piece(s), for k-tile kt0; advanced by `astep` / `bstep` per k-tile -------------
    unsigned va[APW], vb[BPW];
    unsigned astep, bstep;
    if constexpr (!TN) {
        astep = bstep = BX_BK * 2;
#pragma unroll
        for (int u = 0; u < APW; ++u) {
            const int r = (wave * APW + u) * 16 + (lane >> 2);                  // row of the tile image
            const int cs = (lane & 3) ^ ((r >> 2) & 3);                         // source chunk for LDS chunk (lane & 3)
            va[u] = ((unsigned)(m0 + r) * (unsigned)p.A.ld + (unsigned)(kt0 * BX_BK + cs * 8)) * 2u;
        }
#pragma unroll
        for (int u = 0; u < BPW; ++u) {
            const int r = (wave * BPW + u) * 16 + (lane >> 2);
            const int cs = (lane & 3) ^ ((r >> 2) & 3);
            vb[u] = ((unsigned)(n0 + r) * (unsigned)p.B.ld + (unsigned)(kt0 * BX_BK + cs * 8)) * 2u;
        }
    } else {
        astep = (unsigned)BX_BK * (unsigned)p.A.ld * 2u;
        bstep = (unsigned)BX_BK * (unsigned)p.B.ld * 2u;
#pragma unroll
        for (int u = 0; u < APW; ++u) {
            const int byte = (wave * APW + u) * 1024 + lane * 16;               // position in the [32 k][BM] image
            const int kr = byte / (BM * 2), cp = (byte % (BM * 2)) >> 4;
            const int cs = cp ^ ((kr & 3) << 2);
            va[u] = ((unsigned)(kt0 * BX_BK + kr) * (unsigned)p.A.ld + (unsigned)(m0 + cs * 8)) * 2u;
        }
#pragma unroll
        for (int u = 0; u < BPW; ++u) {
            const int byte = (wave * BPW + u) * 1024 + lane * 16;
            const int kr = byte / (BN * 2), cp = (byte % (BN * 2)) >> 4;
            const int cs = cp ^ ((kr & 3) << 2);
            vb[u] = ((unsigned)(kt0 * BX_BK + kr) * (unsigned)p.B.ld + (unsigned)(n0 + cs * 8)) * 2u;
        }
    }
    // ---- fragment read offsets (bytes inside a plane image) ----------------------------------------------------------------------
    // NT: lane (i = lane & 31, kg = lane >> 5) reads the 8 k of chunk 2 s + kg of row (64 w + 32 t + i)
    // TN: 16-lane block b = lane >> 4, c = lane & 15: the block reads [4 k][16 rows]; the lane supplies the address of k-row
    //     (c >> 2), rows 4 (c & 3) .. + 3 of the block and receives the 4 k of row c: blocks 0 / 1 = rows 0-15 / 16-31 at
    //     k 0-7, blocks 2 / 3 the same rows at k 8-15 (operand layout of the 32 x 32 x 16 MFMA)
    int fa[2], fb[2];                                  // NT: per k16-step s; TN: per 32-row tile t
    if constexpr (!TN) {
        const int i = lane & 31, kg = lane >> 5, x = (i >> 2) & 3;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            fa[s] = (wm * 64 + i) * 64 + (((2 * s + kg) ^ x) << 4);
            fb[s] = (wn * 64 + i) * 64 + (((2 * s + kg) ^ x) << 4);
        }
    } else {
        const int b = lane >> 4, c = lane & 15, r = c >> 2;
        const int cb = 2 * (b & 1) + ((c & 3) >> 1);
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            fa[t] = (8 * (b >> 1) + r) * (BM * 2) + ((((wm * 8 + 4 * t) ^ (r << 2)) + cb) << 4) + 8 * (c & 1);
            fb[t] = (8 * (b >> 1) + r) * (BN * 2) + ((((wn * 8 + 4 * t) ^ (r << 2)) + cb) << 4) + 8 * (c & 1);
        }
    }
    auto frag = [&](const unsigned char* plane, int t, int s, bool is_a) __attribute__((always_inline)) -> bx_bf16x8 {
        if constexpr (!TN) {
            const int off = (is_a ? fa[s] : fb[s]) + t * 32 * 64;
            return __builtin_bit_cast(bx_bf16x8, *reinterpret_cast<const bx_u32x4*>(plane + off));
        } else {
            const int rs = (is_a ? BM : BN) * 2;
            const int off = (is_a ? fa[t] : fb[t]) + 16 * s * rs;
            const bx_s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((bx_s16x4 __attribute__((address_space(3)))*)(plane + off));
            const bx_s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((bx_s16x4 __attribute__((address_space(3)))*)(plane + off + 4 * rs));
            typedef short s16x8 __attribute__((ext_vector_type(8)));
            const s16x8 v = (s16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            return __builtin_bit_cast(bx_bf16x8, v);
        }
    };

    bx_f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // ---- software pipeline -------------------------------------------------------------------------------------------------------
    // Two fragment sets: while the 8 NP MFMAs of one 16-k step run on one set, the ds_reads of the next step fill the other, and
    // the DMA instructions of k-tile kt + NS - 1 are issued BETWEEN the MFMAs (a DMA instruction holds the wave's issue port
    // for 60-180 cycles, MI355X_MICROARCH.md: twelve of them in a block in front of the MFMAs left the matrix pipes idle 60 % of
    // the time -- profiles/r04_bx3_sq_v0.txt).  The loop body has no branch: DMA requests beyond the last k-tile are sent with an
    // out-of-range offset (the descriptor's range check answers with zeros, no memory traffic), so the counted vmcnt is the
    // same in every iteration.  Order per k-tile:
    //   phase A   reads (kt, s = 1) -> set 1 | DMA (kt + NS - 1) | MFMAs on set 0 = (kt, s = 0)
    //   vmcnt: k-tile kt + 1 has landed for this wave; lgkmcnt(0): this wave's reads of tile kt are complete; barrier
    //   phase B   reads (kt + 1, s = 0) -> set 0 | MFMAs on set 1 = (kt, s = 1)
    const int nk = kt1 - kt0;
    const unsigned OOB = 0xFFFFFF00u;                  // beyond every num_records (bx3_ok keeps planes below 4e9 bytes)
    bx_bf16x8 a0[2][NP], b0[2][NP], a1[2][NP], b1[2][NP];
    auto load_set = [&](bx_bf16x8 (&fa_)[2][NP], bx_bf16x8 (&fb_)[2][NP], int st, int s) __attribute__((always_inline)) {
        if constexpr (DBG == 1) return;
        const unsigned char* sa = bx_smem + st * Cf::STAGE;
        const unsigned char* sbp = sa + 3 * Cf::APL;
#pragma unroll
        for (int q = 0; q < NP; ++q)
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                fa_[t][q] = frag(sa + q * Cf::APL, t, s, true);
                fb_[t][q] = frag(sbp + q * Cf::BPL, t, s, false);
            }
    };
    auto mma_set = [&](const bx_bf16x8 (&af)[2][NP], const bx_bf16x8 (&bf)[2][NP]) __attribute__((always_inline)) {
        // six piece products per output tile, smallest first; consecutive MFMAs go to different accumulators
#define EAGCN_BX_PROD(PA, PB)                                                                                          \
    _Pragma("unroll") for (int i = 0; i < 2; ++i) _Pragma("unroll") for (int j = 0; j < 2; ++j)                       \
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][PA], bf[j][PB], acc[i][j], 0, 0, 0);
        if constexpr (DBG == 1) return;                // (probe: fill only)
        if constexpr (NP == 3) {
            EAGCN_BX_PROD(2, 0)
            EAGCN_BX_PROD(1, 1)
            EAGCN_BX_PROD(0, 2)
            EAGCN_BX_PROD(1, 0)
            EAGCN_BX_PROD(0, 1)
        }
        EAGCN_BX_PROD(0, 0)
#undef EAGCN_BX_PROD
    };
    auto issue_or_skip = [&](int stage, bool real) __attribute__((always_inline)) {
        if constexpr (DBG == 2) return;                // (probe: compute only, on whatever the LDS holds)
        unsigned char* sb = bx_smem + stage * Cf::STAGE;
#pragma unroll
        for (int q = 0; q < NP; ++q) {
#pragma unroll
            for (int u = 0; u < APW; ++u)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(ra[q], (bx_lds_ptr)(sb + q * Cf::APL + (wave * APW + u) * 1024), 16,
                                                         (int)(real ? va[u] : OOB), 0, 0, 0);
#pragma unroll
            for (int u = 0; u < BPW; ++u)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rb[q], (bx_lds_ptr)(sb + 3 * Cf::APL + q * Cf::BPL + (wave * BPW + u) * 1024), 16,
                                                         (int)(real ? vb[u] : OOB), 0, 0, 0);
        }
#pragma unroll
        for (int u = 0; u < APW; ++u) va[u] += astep;
#pragma unroll
        for (int u = 0; u < BPW; ++u) vb[u] += bstep;
    };
    // prologue: NS - 1 tiles requested (tiles beyond nk as zero fills), tile 0 awaited, set 0 = (0, s = 0)
#pragma unroll
    for (int s = 0; s < NS - 1; ++s) issue_or_skip(s, s < nk);
    bx_wait_vm<PPT * (NS - 2)>();
    __builtin_amdgcn_s_barrier();
    load_set(a0, b0, 0, 0);
    int stage = 0;
    // the last k-tile of an NT product may hold only 16 valid k (K is a multiple of 16): its second half is skipped
    const bool half_tail = !TN && (kt1 * BX_BK > Kx);
    for (int kt = 0; kt < nk; ++kt) {
        int st_next = stage + 1;
        if (st_next == NS) st_next = 0;
        int st_dma = stage + NS - 1;
        if (st_dma >= NS) st_dma -= NS;
        // ---- phase A
        load_set(a1, b1, stage, 1);
        issue_or_skip(st_dma, kt + NS - 1 < nk);
        mma_set(a0, b0);
#pragma unroll
        for (int g = 0; g < 4 * NP; ++g) {
            __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
        }
        // ---- k-tile kt + 1 is complete for every wave; nobody reads tile kt's stage after this barrier
        __builtin_amdgcn_sched_barrier(0);             // (nothing of phase A sinks below: MFMAs are not memory operations)
        bx_wait_vm<PPT * (NS - 2)>();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        // ---- phase B
        load_set(a0, b0, st_next, 0);
        if (!(half_tail && kt == nk - 1)) mma_set(a1, b1);
#pragma unroll
        for (int g = 0; g < 4 * NP; ++g) {
            __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        stage = st_next;
    }
    // the zero-fill requests behind the last k-tile have landed and every wave is done with the LDS stages before the next unit's
    // prologue overwrites them
    bx_wait_vm<0>();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    // ---- epilogue: D layout col = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5) ------------------------------------------
    const int Mlim = TN ? p.M : Mx;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int col = n0 + wn * 64 + j * 32 + (lane & 31);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (row < Mlim && col < p.N) Cz[(size_t)row * p.ldc + col] = acc[i][j][r];
            }
        }
}

__device__ __forceinline__ int bx_mx(const BxProb& p) { return (!p.tn && p.M_dev) ? min(*p.M_dev, p.M) : p.M; }
__device__ __forceinline__ int bx_kx(const BxProb& p) { return (p.tn && p.K_dev) ? min(*p.K_dev, p.K) : p.K; }

// work units of one problem: NT: output tiles; TN: (tile, k-chunk) items, chunk z -> slab z
struct BxUnits { int tiles_m, tiles_n, splits, kt_total, kt_per; int n; };
__device__ __forceinline__ BxUnits bx_units(const BxProb& p, int Mx, int Kx, int BM, int BN) {
    BxUnits u;
    u.tiles_m = (Mx + BM - 1) / BM;
    u.tiles_n = (p.N + BN - 1) / BN;
    u.kt_total = max(1, (Kx + BX_BK - 1) / BX_BK);
    u.splits = p.tn ? max(1, p.splits) : 1;
    u.kt_per = (u.kt_total + u.splits - 1) / u.splits;
    u.n = u.tiles_m * u.tiles_n * u.splits;
    return u;
}

template <int WM, int WN, int NP, int DBG>
__device__ __forceinline__ void bx_run_unit(const BxProb& p, const BxUnits& u, int Mx, int Kx, int id) {
    // unit order: NT: column tile fastest (the tiles of a row panel are neighbours); TN: k-chunk MAJOR -- the tiles of one k-chunk
    // are neighbours and share its rows of both operands in their XCD's L2 (tile-major order measured a 13 % L2 hit rate: every
    // byte of X and dP crossed the fabric once per tile, profiles/r04_bx3_tcc_v1.txt)
    if (!p.tn) {
        const int tm = id / u.tiles_n, tn = id - tm * u.tiles_n;
        bx_tile<false, WM, WN, NP, DBG>(p, Mx, Kx, tm, tn, 0, u.kt_total, p.C);
    } else {
        const int tiles = u.tiles_m * u.tiles_n;
        const int z = id / tiles, tile = id - z * tiles;
        const int tm = tile / u.tiles_n, tn = tile - tm * u.tiles_n;
        const int kt0 = min(z * u.kt_per, u.kt_total), kt1 = min(kt0 + u.kt_per, u.kt_total);
        bx_tile<true, WM, WN, NP, DBG>(p, Mx, Kx, tm, tn, kt0, kt1, p.C + (size_t)z * p.slab);       // (an empty chunk stores zeros)
    }
}

// up to two problems in one persistent launch (the dX / dW pair of a layer's backward)
template <int WM, int WN, int NP, int DBG = 0>
__global__ __launch_bounds__(64 * WM * WN, (WM * WN) / 4) void bx3_kernel(BxProb p0, BxProb p1, int has1) {
    using Cf = BxCfg<WM, WN>;
    const int G = gridDim.x;
    const int Mx0 = bx_mx(p0), Kx0 = bx_kx(p0);
    const BxUnits u0 = bx_units(p0, Mx0, Kx0, Cf::BM, Cf::BN);
    int Mx1 = 0, Kx1 = 0;
    BxUnits u1;
    u1.n = 0;
    if (has1) {
        Mx1 = bx_mx(p1);
        Kx1 = bx_kx(p1);
        u1 = bx_units(p1, Mx1, Kx1, Cf::BM, Cf::BN);
    }
    const int total = u0.n + u1.n;
    // XCD x (dispatch slot b runs on XCD b % 8) owns the contiguous units [x chunk, (x + 1) chunk); its G / 8 workgroups take them
    // round-robin
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, per = max(1, G >> 3);
    const int chunk = (total + 7) >> 3;
    const int lo = xcd * chunk, hi = min(total, lo + chunk);
    for (int id = lo + slot; id < hi; id += per) {
        // (ONE instantiation of each form: the problem is selected by value, not by a second copy of the tile code)
        const bool second = id >= u0.n;
        const BxProb& p = second ? p1 : p0;
        const BxUnits& u = second ? u1 : u0;
        bx_run_unit<WM, WN, NP, DBG>(p, u, second ? Mx1 : Mx0, second ? Kx1 : Kx0, second ? id - u0.n : id);
    }
}

// ---- operand producers -----------------------------------------------------------------------------------------------------------
// fp32 matrix [rows][ld] -> three bf16 planes (stand-alone conversion: C-ABI entry, tests, layer-level path; the model engine's
// producers write the planes in their own epilogues)
__global__ __launch_bounds__(256) void bx3_split_kernel(const float* __restrict__ x, int rows, const int* __restrict__ rows_dev, int ld,
                                                       uint16_t* __restrict__ pl, size_t pstride, int np) {
    const int R = rows_dev ? min(*rows_dev, rows) : rows;
    const size_t n4 = (size_t)R * ld / 4;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        const float4 v = *reinterpret_cast<const float4*>(x + 4 * i);
        if (np == 3) bx3_store4(pl, pstride, 4 * i, v);
        else bx1_store4(pl, 4 * i, v);
    }
}

int bx3_grid() {
    static const int g = [] {
        int cus = 256, dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
            cus = prop.multiProcessorCount;
        (void)hipGetLastError();
        const char* e = getenv("EAGCN_BX3_WGS");
        int v = e ? atoi(e) : cus;
        return std::max(8, v / 8 * 8);
    }();
    return g;
}

int launch_bx3_split(const float* x, int rows, const int* rows_dev, int ld, uint16_t* planes, size_t pstride, int np, hipStream_t s) {
    EAGCN_CHECK_ARG(x && planes && (ld & 3) == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0 && (reinterpret_cast<uintptr_t>(planes) & 7) == 0 &&
                        (pstride & 3) == 0, "bx3_split: operands must be 16-byte aligned with ld a multiple of 4");
    if (rows <= 0) return EAGCN_OK;
    const size_t n4 = (size_t)rows * ld / 4;
    bx3_split_kernel<<<(unsigned)std::max<size_t>(1, std::min<size_t>((n4 + 255) / 256, 4096)), 256, 0, s>>>(x, rows, rows_dev, ld, planes, pstride, np);
    EAGCN_LAUNCH_CHECK();
    return EAGCN_OK;
}

bool bx3_ok(const BxProb& p) {
    if (p.M <= 0 || p.N <= 0 || p.K <= 0 || !p.A.p || !p.B.p || !p.C) return false;
    if ((p.A.ld & 7) || (p.B.ld & 7) || (reinterpret_cast<uintptr_t>(p.A.p) & 15) || (reinterpret_cast<uintptr_t>(p.B.p) & 15)) return false;
    if ((p.A.pstride & 7) || (p.B.pstride & 7)) return false;
    // 32-bit byte offsets inside a plane (buffer descriptor + voffset)
    const double arows = p.tn ? p.K : p.M, brows = p.tn ? p.K : p.N;
    if ((arows + 256) * p.A.ld * 2.0 >= 4.0e9 || (brows + 256) * p.B.ld * 2.0 >= 4.0e9) return false;
    if (!p.tn) return (p.K & 15) == 0 && p.K <= p.A.ld && p.K <= p.B.ld && !p.K_dev;
    return !p.M_dev && p.M <= p.A.ld && p.N <= p.B.ld && p.splits >= 1 && (p.splits == 1 || p.slab >= (size_t)p.M * p.ldc);
}

template <int WM, int WN, int NP, int DBG = 0>
static int bx3_launch_cfg(const BxProb& p0, const BxProb* p1, hipStream_t s) {
    using Cf = BxCfg<WM, WN>;
    constexpr int lds = Cf::NS * Cf::STAGE;
    static bool attr_done = false;
    if (!attr_done) {
        EAGCN_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&bx3_kernel<WM, WN, NP, DBG>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        attr_done = true;
    }
    bx3_kernel<WM, WN, NP, DBG><<<bx3_grid(), 64 * WM * WN, lds, s>>>(p0, p1 ? *p1 : p0, p1 ? 1 : 0);
    EAGCN_LAUNCH_CHECK();
    return EAGCN_OK;
}

// np = 3: exact products; np = 1: plain bf16 operands (one plane, one product)
int launch_bx3(const BxProb& p0, const BxProb* p1, int np, hipStream_t s, double work, int prof_tag) {
    EAGCN_CHECK_ARG(bx3_ok(p0) && (!p1 || bx3_ok(*p1)), "bx3 gemm: operands not aligned / extents unsupported");
    EAGCN_CHECK_ARG(np == 1 || np == 3, "bx3 gemm: 1 or 3 planes");
    ProfScope ps(prof_tag, s, work);
    // tile choice: the 8-wave 256 x 128 tile halves the L2 -> LDS traffic per flop but needs enough row panels to fill the chip
    static const int forced = [] { const char* e = getenv("EAGCN_BX3_TILE"); return e ? atoi(e) : -1; }();
    int big = forced;
    if (big < 0) {
        const long rows = p0.tn ? p0.M : p0.M;       // (capacity; the kernel counts its units from the device-side extents)
        big = 0;
        (void)rows;
    }
    static const int dbg = [] { const char* e = getenv("EAGCN_BX3_DBG"); return e ? atoi(e) : 0; }();     // probes (wrong results!)
    if (np == 3 && dbg == 1) return bx3_launch_cfg<2, 2, 3, 1>(p0, p1, s);
    if (np == 3 && dbg == 2) return bx3_launch_cfg<2, 2, 3, 2>(p0, p1, s);
    if (np == 3) return big == 1 ? bx3_launch_cfg<4, 2, 3>(p0, p1, s) : bx3_launch_cfg<2, 2, 3>(p0, p1, s);
    return big == 1 ? bx3_launch_cfg<4, 2, 1>(p0, p1, s) : bx3_launch_cfg<2, 2, 1>(p0, p1, s);
}

}  // namespace eagcn

using namespace eagcn;

/* planes of a row-major fp32 matrix: three bf16 planes (np = 3: x = x0 + x1 + x2 exactly) or one (np = 1: round to nearest even),
 * plane q at planes + q * plane_stride (elements); reference-free helper of the C-ABI product below and of tests */
extern "C" int eagcn_bx3_split(const float* x, int rows, int ld, uint16_t* planes, size_t plane_stride, int np, void* stream) {
    EAGCN_CHECK_ARG(np == 1 || np == 3, "eagcn_bx3_split: np must be 1 or 3");
    return launch_bx3_split(x, rows, nullptr, ld, planes, plane_stride, np, (hipStream_t)stream);
}

/* C = op(A) . op(B) from bf16 planes.  tn = 0: C[M,N] = A[M,K] . B[N,K]^T (A planes [M][lda], B planes [N][ldb]);
 * tn = 1: C[M,N] = A[K,M]^T . B[K,N] (planes [K][lda], [K][ldb]) written as `splits` partial slabs C + z * slab (their sum is the
 * product).  Optional second problem in the same launch (has1). */
extern "C" int eagcn_gemm_bx3(int tn, int M, int N, int K, const uint16_t* A, size_t a_pstride, int lda, const uint16_t* B,
                              size_t b_pstride, int ldb, float* C, int ldc, int splits, size_t slab, int np, void* stream) {
    BxProb p;
    memset(&p, 0, sizeof(p));
    p.A = BxPlanes{A, a_pstride, lda}; p.B = BxPlanes{B, b_pstride, ldb}; p.C = C; p.ldc = ldc;
    p.M = M; p.N = N; p.K = K; p.tn = tn; p.splits = tn ? std::max(1, splits) : 1; p.slab = slab;
    return launch_bx3(p, nullptr, np, (hipStream_t)stream, 2.0 * M * N * K, PROF_GEMM);
}

/* the dX / dW pair of a layer's backward in ONE persistent launch: problem 0 NT, problem 1 TN with split-K slabs */
extern "C" int eagcn_gemm_bx3_pair(int M0, int N0, int K0, const uint16_t* A0, size_t a0_pstride, int lda0, const uint16_t* B0,
                                   size_t b0_pstride, int ldb0, float* C0, int ldc0, int M1, int N1, int K1, const uint16_t* A1,
                                   size_t a1_pstride, int lda1, const uint16_t* B1, size_t b1_pstride, int ldb1, float* C1, int ldc1,
                                   int splits, size_t slab, int np, void* stream) {
    BxProb p, q;
    memset(&p, 0, sizeof(p));
    memset(&q, 0, sizeof(q));
    p.A = BxPlanes{A0, a0_pstride, lda0}; p.B = BxPlanes{B0, b0_pstride, ldb0}; p.C = C0; p.ldc = ldc0;
    p.M = M0; p.N = N0; p.K = K0; p.tn = 0; p.splits = 1;
    q.A = BxPlanes{A1, a1_pstride, lda1}; q.B = BxPlanes{B1, b1_pstride, ldb1}; q.C = C1; q.ldc = ldc1;
    q.M = M1; q.N = N1; q.K = K1; q.tn = 1; q.splits = std::max(1, splits); q.slab = slab;
    return launch_bx3(p, &q, np, (hipStream_t)stream, 2.0 * M0 * N0 * K0 + 2.0 * M1 * N1 * K1, PROF_GEMM_PAIR);
}
