// fp32-EQUIVALENT layer products at the bf16 matrix rate (exact operand split; piece products to 3 x 2^-24 |a b|; fp32 accumulate): operands arrive as THREE bf16 PLANES written by their producers.
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
// Operand planes are PANEL-MAJOR (bx3.h): 32-column panels of all rows, 64 bytes per row, the four 16-byte chunks of a row
// XOR-swizzled by (row >> 2) & 3 -- the global image IS the LDS image, every LDS-DMA instruction copies 1 KB of consecutive bytes:
//   NT  operand tile = 128 rows of ONE panel (k-tile kt = panel kt): 8 KB contiguous; LDS [128 rows][64 B]
//   TN  operand tile = 32 k-rows of FOUR panels (128 columns):       4 x 2 KB;        LDS [4 panels][32 k-rows][64 B]
// (round-4 first version: row-major planes, a k-contiguous tile was 128 x 64-byte segments and the fill ran at 15-20 B/clk/CU:
// forward 73 us, pair 160 us at 19 200 rows; profiles/r04_lds_fill_probe.txt.)
// Fragment reads: NT ds_read_b128 of chunk (2 s + kg) ^ ((row >> 2) & 3); TN ds_read_b64_tr_b16 (the transposing read hands a lane
// the four k-consecutive values of ITS column out of a [4 k][16 column] block -- tools/probes/bx3_probe.hip pins the mapping): a
// 16-lane block touches 4 k-rows x 32 bytes, a half wave 4 rows x 64 bytes = 256 distinct bytes.
// What is NOT inside the actual extents: k beyond K (NT: chunks of the last panel; TN: rows beyond the device-side row count) is
// requested out of range per lane and arrives as ZERO; rows / columns beyond M / N hold whatever the image holds there -- they
// only reach output rows / columns that are never stored.
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
constexpr int BX_NC = 4;                     // compute waves: 2 x 2, one 64 x 64 accumulator block each (one per SIMD)
constexpr int BX_NL = 2;                     // loader waves (four measured the same, with row-major planes and with the panel-major images: DESIGN.md section 4)
constexpr int BX_BM = BX3_BM, BX_BN = BX3_BN;
constexpr int BX_APL = BX_BM * 64, BX_BPL = BX_BN * 64;          // bytes of one plane image of a k-tile
constexpr int BX_STAGE = 3 * (BX_APL + BX_BPL);                  // 48 KB
constexpr int BX_NS = 3;                                         // LDS stages (144 KB)
constexpr int BX_APW = BX_BM / 16 / BX_NL, BX_BPW = BX_BN / 16 / BX_NL;   // 1 KB DMA pieces per loader wave and plane

extern __shared__ __attribute__((aligned(1024))) unsigned char bx_smem[];

template <int N>
__device__ __forceinline__ void bx_wait_vm() {
    static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit counter");
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// ---- work units --------------------------------------------------------------------------------------------------------------------
// NT: a unit is an output tile (all its k-tiles); TN: a (tile, k-chunk) item, chunk z -> slab z.  Unit order: NT: column tile
// fastest (the tiles of a row panel are neighbours); TN: k-chunk MAJOR -- the tiles of one k-chunk are neighbours and share its
// rows of both operands in their XCD's L2 (tile-major order measured a 13 % L2 hit rate: every byte of X and dP crossed the
// fabric once per tile, profiles/r04_bx3_tcc_v1.txt).
struct BxUnits { int tiles_m, tiles_n, splits, kt_total, kt_per; int n; };
__device__ __forceinline__ int bx_mx(const BxProb& p) { return (!p.tn && p.M_dev) ? min(*p.M_dev, p.M) : p.M; }
__device__ __forceinline__ int bx_kx(const BxProb& p) { return (p.tn && p.K_dev) ? min(*p.K_dev, p.K) : p.K; }
__device__ __forceinline__ BxUnits bx_units(const BxProb& p, int Mx, int Kx, int ou = 0, int okt = 0) {
    BxUnits u;
    u.tiles_m = (Mx + BX_BM - 1) / BX_BM;
    u.tiles_n = (p.N + BX_BN - 1) / BX_BN;
    u.kt_total = max(1, (Kx + BX_BK - 1) / BX_BK);
    // TN: the k-tiles are spread over bx3_used_splits() chunks (bx3.h; capacity-sized launches: the actual K may be a fraction of the
    // capacity the host sized `splits` for); only those slabs are written, and the consumer sums only those
    u.splits = p.tn ? bx3_used_splits_wave(max(1, p.splits), Kx, u.tiles_m * u.tiles_n, ou, okt, max(1, (int)gridDim.x >> 3)) : 1;
    u.kt_per = (u.kt_total + u.splits - 1) / u.splits;
    u.n = u.tiles_m * u.tiles_n * u.splits;
    return u;
}
// the units of ONE problem that this workgroup owns: ids first, first + step, ... (count of them)
struct BxStream { BxUnits u; int Mx, Kx, first, step, count; };
struct BxUnit { int tm, tn, kt0, nk, z; };
// (__host__ __device__: it is called from a lambda, which clang treats as host + device code in its host pass)
__host__ __device__ __forceinline__ BxUnit bx_unit(const BxProb& p, const BxStream& st, int j) {
    const int id = st.first + j * st.step;
    BxUnit r;
    if (!p.tn) {
        r.tm = id / st.u.tiles_n; r.tn = id - r.tm * st.u.tiles_n; r.kt0 = 0; r.nk = st.u.kt_total; r.z = 0;
    } else {
        const int tiles = st.u.tiles_m * st.u.tiles_n;
        r.z = id / tiles;
        const int tile = id - r.z * tiles;
        r.tm = tile / st.u.tiles_n; r.tn = tile - r.tm * st.u.tiles_n;
        const int k0 = r.z * st.u.kt_per, k1 = k0 + st.u.kt_per;
        r.kt0 = k0 < st.u.kt_total ? k0 : st.u.kt_total;
        r.nk = (k1 < st.u.kt_total ? k1 : st.u.kt_total) - r.kt0;
    }
    return r;
}
__device__ __forceinline__ int bx_total_tiles(const BxProb& p, const BxStream& st) {
    if (!p.tn) return st.count * st.u.kt_total;
    int t = 0;
    for (int j = 0; j < st.count; ++j) t += bx_unit(p, st, j).nk;
    return t;
}

// ---- loader waves ------------------------------------------------------------------------------------------------------------------
// Two waves do nothing but LDS-DMA: wave lw requests every second 1 KB piece of the three (NP) planes of both operands of a
// k-tile -- 8 NP instructions -- for k-tile k + 2 while the compute waves multiply k-tile k.  (Issued by the compute waves
// themselves the twelve requests per wave and k-tile held the wave's issue port while the matrix pipe idled: fill alone 65 us,
// products alone 59 us, together 99 us at 19 200 x 400 x 720 -- DESIGN.md section 4 (probes EAGCN_BX3_DBG=1 / 2).)
template <bool TN, int NP>
__device__ __forceinline__ void bx_loader(const BxProb& p, const BxStream& st, const int lw) {
    constexpr int PPL = NP * (BX_APW + BX_BPW);
    const int lane = threadIdx.x & 63;
    // buffer descriptors: one per operand plane over the whole image
    const unsigned apanel = (unsigned)p.A.rows * 64u, bpanel = (unsigned)p.B.rows * 64u;      // bytes of one 32-column panel
    const unsigned abytes = (unsigned)((p.A.ld + 31) >> 5) * apanel, bbytes = (unsigned)((p.B.ld + 31) >> 5) * bpanel;
    __amdgpu_buffer_rsrc_t ra[3], rb[3];           // (fixed size: clang's host pass rejects the DMA builtin on an array of dependent extent)
#pragma unroll
    for (int q = 0; q < NP; ++q) {
        ra[q] = __builtin_amdgcn_make_buffer_rsrc((void*)(p.A.p + (size_t)q * p.A.pstride), 0, (int)abytes, 0x00020000);
        rb[q] = __builtin_amdgcn_make_buffer_rsrc((void*)(p.B.p + (size_t)q * p.B.pstride), 0, (int)bbytes, 0x00020000);
    }
    // per-lane source offsets (bytes) RELATIVE to the k-tile's origin: piece pc of an operand's tile image is 1 KB of consecutive
    // bytes (NT: rows 16 pc .. of the k-tile's panel; TN: rows 16 (pc & 1) .. of the tile's panel pc >> 1), lane-linear
    unsigned ra_[BX_APW], rb_[BX_BPW];
    const unsigned astep = TN ? BX_BK * 64u : apanel;                  // k-tile to k-tile
    const unsigned bstep = TN ? BX_BK * 64u : bpanel;
#pragma unroll
    for (int u = 0; u < BX_APW; ++u) {
        const int pc = lw + BX_NL * u;
        ra_[u] = TN ? (unsigned)(pc >> 1) * apanel + (unsigned)((pc & 1) * 1024 + lane * 16) : (unsigned)(pc * 1024 + lane * 16);
    }
#pragma unroll
    for (int u = 0; u < BX_BPW; ++u) {
        const int pc = lw + BX_NL * u;
        rb_[u] = TN ? (unsigned)(pc >> 1) * bpanel + (unsigned)((pc & 1) * 1024 + lane * 16) : (unsigned)(pc * 1024 + lane * 16);
    }
    const unsigned OOB = 0xFFFFFF00u;                  // beyond every num_records (bx3_ok keeps the images below 4e9 bytes)
    // k beyond the actual K arrives as zeros (requested out of range), and the compute waves multiply every k-tile in full:
    //   NT: the last panel may hold fewer than 32 valid k (K is a multiple of 8): the lane's LOGICAL chunk decides
    //   TN: k-rows beyond the (device-side) row count: the lane's row decides
    int kca[BX_APW], kcb[BX_BPW];                      // k offset of this lane's 16 bytes inside a k-tile
#pragma unroll
    for (int u = 0; u < BX_APW; ++u) {
        const int pc = lw + BX_NL * u, r = pc * 16 + (lane >> 2);
        kca[u] = TN ? (pc & 1) * 16 + (lane >> 2) : 8 * ((lane & 3) ^ ((r >> 2) & 3));
    }
#pragma unroll
    for (int u = 0; u < BX_BPW; ++u) {
        const int pc = lw + BX_NL * u, r = pc * 16 + (lane >> 2);
        kcb[u] = TN ? (pc & 1) * 16 + (lane >> 2) : 8 * ((lane & 3) ^ ((r >> 2) & 3));
    }
    // iterator over the k-tiles of the stream
    int j = -1, left = 0, kpos = 0;                    // kpos: first k of the current k-tile
    unsigned oa = 0, ob = 0;                           // byte offsets of the current k-tile's origin in A / B
    auto issue_next = [&](int stage) __attribute__((always_inline)) {
        while (left == 0 && j + 1 < st.count) {        // next unit that has k-tiles
            ++j;
            const BxUnit un = bx_unit(p, st, j);
            left = un.nk;
            kpos = un.kt0 * BX_BK;
            if constexpr (!TN) {                       // panel kt0, rows of the tile
                oa = (unsigned)un.kt0 * apanel + (unsigned)(un.tm * BX_BM) * 64u;
                ob = (unsigned)un.kt0 * bpanel + (unsigned)(un.tn * BX_BN) * 64u;
            } else {                                   // the tile's first panel, k-row kt0 * 32
                oa = (unsigned)(un.tm * (BX_BM / 32)) * apanel + (unsigned)(un.kt0 * BX_BK) * 64u;
                ob = (unsigned)(un.tn * (BX_BN / 32)) * bpanel + (unsigned)(un.kt0 * BX_BK) * 64u;
            }
        }
        const bool real = left > 0;                    // behind the last k-tile: zero fills (out-of-range requests, no traffic) keep
        unsigned char* sb = bx_smem + stage * BX_STAGE;   // the counted vmcnt uniform
#pragma unroll
        for (int q = 0; q < NP; ++q) {
#pragma unroll
            for (int u = 0; u < BX_APW; ++u)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(ra[q], (bx_lds_ptr)(sb + q * BX_APL + (lw + BX_NL * u) * 1024), 16,
                                                         (int)((real && kpos + kca[u] < st.Kx) ? oa + ra_[u] : OOB), 0, 0, 0);
#pragma unroll
            for (int u = 0; u < BX_BPW; ++u)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rb[q], (bx_lds_ptr)(sb + 3 * BX_APL + q * BX_BPL + (lw + BX_NL * u) * 1024), 16,
                                                         (int)((real && kpos + kcb[u] < st.Kx) ? ob + rb_[u] : OOB), 0, 0, 0);
        }
        if (real) { oa += astep; ob += bstep; kpos += BX_BK; --left; }
    };
    const int total = bx_total_tiles(p, st);
    issue_next(0);
    issue_next(1);
    bx_wait_vm<PPL>();                                 // k-tile 0 has landed (this wave's share of it)
    __builtin_amdgcn_s_barrier();                      // B_P
    int stage = 2;
    for (int k = 0; k < total; ++k) {
        issue_next(stage);                             // k-tile k + 2 -> the stage nobody has read since barrier k - 1
        if (++stage == BX_NS) stage = 0;
        bx_wait_vm<PPL>();                             // k-tile k + 1 has landed
        __builtin_amdgcn_s_barrier();                  // B_k
    }
    bx_wait_vm<0>();                                   // the zero fills behind the last k-tile
    __builtin_amdgcn_s_barrier();                      // B_end
}

// ---- compute waves -----------------------------------------------------------------------------------------------------------------
// fragment read offsets (bytes inside a plane image):
//   NT: lane (i = lane & 31, kg = lane >> 5) reads the 8 k of chunk 2 s + kg of row (64 w + 32 t + i)
//   TN: 16-lane block b = lane >> 4, c = lane & 15: the block reads [4 k][16 rows]; the lane supplies the address of k-row
//       (c >> 2), rows 4 (c & 3) .. + 3 of the block and receives the 4 k of row c: blocks 0 / 1 = rows 0-15 / 16-31 at
//       k 0-7, blocks 2 / 3 the same rows at k 8-15 (operand layout of the 32 x 32 x 16 MFMA)
template <bool TN, int NP, int DBG>
__device__ __forceinline__ void bx_compute(const BxProb& p, const BxStream& st, const int wave) {
    const int lane = threadIdx.x & 63;
    const int wm = wave >> 1, wn = wave & 1;
    int fa[2], fb[2], fah[2], fbh[2];                  // NT: per k16-step s; TN: per 32-column tile t (lo / hi: k-rows +0 / +4)
    if constexpr (!TN) {
        const int i = lane & 31, kg = lane >> 5, x = (i >> 2) & 3;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            fa[s] = (wm * 64 + i) * 64 + (((2 * s + kg) ^ x) << 4);
            fb[s] = (wn * 64 + i) * 64 + (((2 * s + kg) ^ x) << 4);
        }
    } else {
        // block b, lane c: k-row 8 (b >> 1) + (c >> 2) [+ 4], columns 16 (b & 1) + 4 (c & 3) .. + 3 of the 32-column tile = panel
        // (2 w + t) of the image: logical chunk 2 (b & 1) + ((c & 3) >> 1), swizzled by (k-row >> 2) & 3 = 2 (b >> 1) [+ 1]
        const int b = lane >> 4, c = lane & 15;
        const int row = 8 * (b >> 1) + (c >> 2), ch = 2 * (b & 1) + ((c & 3) >> 1), half = 8 * (c & 1);
        const int lo_off = row * 64 + ((ch ^ (2 * (b >> 1))) << 4) + half;
        const int hi_off = (row + 4) * 64 + ((ch ^ (2 * (b >> 1) + 1)) << 4) + half;
        // (tile t of a wave = panel 2 w + t: + t * 2048 bytes, a compile-time offset of the read -- only FOUR offset registers live
        //  across the k-loop; eight put the 128 x 128 kernel two registers over its 256 and it re-loaded them from scratch every k-tile)
        fa[0] = (2 * wm) * 2048 + lo_off; fah[0] = (2 * wm) * 2048 + hi_off;
        fb[0] = (2 * wn) * 2048 + lo_off; fbh[0] = (2 * wn) * 2048 + hi_off;
        fa[1] = fb[1] = fah[1] = fbh[1] = 0;
    }
    auto frag = [&](const unsigned char* plane, int t, int s, bool is_a) __attribute__((always_inline)) -> bx_bf16x8 {
        if constexpr (!TN) {
            const int off = (is_a ? fa[s] : fb[s]) + t * 32 * 64;
            return __builtin_bit_cast(bx_bf16x8, *reinterpret_cast<const bx_u32x4*>(plane + off));
        } else {
            const int off = (is_a ? fa[0] : fb[0]) + t * 2048 + 16 * s * 64, offh = (is_a ? fah[0] : fbh[0]) + t * 2048 + 16 * s * 64;
            const bx_s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((bx_s16x4 __attribute__((address_space(3)))*)(plane + off));
            const bx_s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((bx_s16x4 __attribute__((address_space(3)))*)(plane + offh));
            typedef short s16x8 __attribute__((ext_vector_type(8)));
            const s16x8 v = (s16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            return __builtin_bit_cast(bx_bf16x8, v);
        }
    };
    // TWO accumulators per output tile.  v_mfma_f32_32x32x16_bf16 does not round its accumulation to nearest: a same-sign
    // reduction through ONE accumulator drifts by about -2.7e-11 K relative (tools/bx3_bench bias: -6e-7 at K = 25 000, -3e-6 at
    // 100 000; the fp32 MFMA stays at 1e-9), because every one of the six piece products of a k-step costs the running sum a
    // truncation at ITS magnitude.  Five of the six carry a weight of 2^-8 or less: they go to `lo`, whose own magnitude is 2^-8
    // of the result (its truncations are worth 2^-32 of the result), and `hi` sees ONE instruction per k-step -- the exact
    // products a0.b0 -- i.e. a sixth of the truncations and of the drift (about -2e-8 over the 4096-row k-chunks the host
    // cuts a weight gradient into).  hi + lo (v_add_f32: round to nearest even) is formed once, in the epilogue.
    bx_f32x16 acc[2][2], lo[2][2];
    auto zero_acc = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int jj = 0; jj < 2; ++jj)
#pragma unroll
                for (int r = 0; r < 16; ++r) { acc[i][jj][r] = 0.f; lo[i][jj][r] = 0.f; }
    };
    bx_bf16x8 a0[2][NP], b0[2][NP], a1[2][NP], b1[2][NP];
    auto load_set = [&](bx_bf16x8 (&fa_)[2][NP], bx_bf16x8 (&fb_)[2][NP], int stg, int s) __attribute__((always_inline)) {
        if constexpr (DBG == 1) return;
        const unsigned char* sa = bx_smem + stg * BX_STAGE;
        const unsigned char* sbp = sa + 3 * BX_APL;
#pragma unroll
        for (int q = 0; q < NP; ++q)
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                fa_[t][q] = frag(sa + q * BX_APL, t, s, true);
                fb_[t][q] = frag(sbp + q * BX_BPL, t, s, false);
            }
    };
    auto mma_set = [&](const bx_bf16x8 (&af)[2][NP], const bx_bf16x8 (&bf)[2][NP]) __attribute__((always_inline)) {
        // six piece products per output tile, smallest first; consecutive MFMAs go to different accumulators
#define EAGCN_BX_PROD(ACC, PA, PB)                                                                                     \
    _Pragma("unroll") for (int i = 0; i < 2; ++i) _Pragma("unroll") for (int jj = 0; jj < 2; ++jj)                     \
        ACC[i][jj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][PA], bf[jj][PB], ACC[i][jj], 0, 0, 0);
        if constexpr (DBG == 1) return;                // (probe: fill only)
        if constexpr (NP == 3) {
            EAGCN_BX_PROD(lo, 2, 0)
            EAGCN_BX_PROD(lo, 1, 1)
            EAGCN_BX_PROD(lo, 0, 2)
            EAGCN_BX_PROD(lo, 1, 0)
            EAGCN_BX_PROD(lo, 0, 1)
        }
        EAGCN_BX_PROD(acc, 0, 0)
#undef EAGCN_BX_PROD
    };
    // D layout: col = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)
    auto store_unit = [&](const BxUnit& un) __attribute__((always_inline)) {
        float* __restrict__ Cz = p.C + (size_t)un.z * p.slab;
        const int Mlim = TN ? p.M : st.Mx;
        const int m0 = un.tm * BX_BM + wm * 64, n0 = un.tn * BX_BN + wn * 64;
        if (m0 + 64 <= Mlim && n0 + 64 <= p.N) {       // whole block inside the matrix: no predicates
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int jj = 0; jj < 2; ++jj) {
                    float* cp = Cz + (size_t)(m0 + i * 32 + 4 * (lane >> 5)) * p.ldc + n0 + jj * 32 + (lane & 31);
#pragma unroll
                    for (int r = 0; r < 16; ++r) cp[(size_t)((r & 3) + 8 * (r >> 2)) * p.ldc] = acc[i][jj][r] + lo[i][jj][r];
                }
        } else {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int jj = 0; jj < 2; ++jj) {
                    const int col = n0 + jj * 32 + (lane & 31);
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int row = m0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                        if (row < Mlim && col < p.N) Cz[(size_t)row * p.ldc + col] = acc[i][jj][r] + lo[i][jj][r];
                    }
                }
        }
    };
    // ---- the stream: one barrier per k-tile -------------------------------------------------------------------------------------
    //   phase A   reads (k, s = 1) -> set 1 | MFMAs on set 0 = (k, s = 0)
    //   lgkmcnt(0): this wave's reads of k-tile k are complete; barrier k: k-tile k + 1 has landed, k-tile k's stage is free
    //   phase B   reads (k + 1, s = 0) -> set 0 | MFMAs on set 1 = (k, s = 1)
    // A unit ends with the store of its accumulators while set 0 already holds the next unit's first half and the loader waves
    // keep requesting: no pipeline drain between units.
    const int total = bx_total_tiles(p, st);
    zero_acc();
    int j = 0;
    BxUnit un = st.count > 0 ? bx_unit(p, st, 0) : BxUnit{0, 0, 0, 1 << 30, 0};
    while (j < st.count && un.nk == 0) {               // (empty k-chunks store zeros)
        store_unit(un);
        if (++j < st.count) un = bx_unit(p, st, j);
    }
    int rem = un.nk;
    __builtin_amdgcn_s_barrier();                      // B_P
    load_set(a0, b0, 0, 0);
    int stage = 0;
    for (int k = 0; k < total; ++k) {
        int st_next = stage + 1;
        if (st_next == BX_NS) st_next = 0;
        // ---- phase A
        load_set(a1, b1, stage, 1);
        mma_set(a0, b0);
#pragma unroll
        for (int g = 0; g < 4 * NP; ++g) {
            __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, TN ? 2 : 1, 0);
        }
        __builtin_amdgcn_sched_barrier(0);             // (nothing of phase A sinks below: MFMAs are not memory operations)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                  // B_k
        __builtin_amdgcn_sched_barrier(0);
        // ---- phase B
        load_set(a0, b0, st_next, 0);
        mma_set(a1, b1);
#pragma unroll
        for (int g = 0; g < 4 * NP; ++g) {
            __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, TN ? 2 : 1, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        stage = st_next;
        if (--rem == 0) {                              // the unit is complete
            store_unit(un);
            zero_acc();
            do {
                if (++j < st.count) un = bx_unit(p, st, j); else un.nk = 1 << 30;
                if (j < st.count && un.nk == 0) store_unit(un);
            } while (j < st.count && un.nk == 0);
            rem = un.nk;
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                      // B_end: every wave is done with the LDS stages
}

// XCD x (dispatch slot b runs on XCD b % 8) owns the contiguous units [x n / 8, (x + 1) n / 8) of a problem; its workgroups take
// them round-robin, starting at slot `rot` (the workgroups that drew one unit more of the first problem draw one less of the second)
__device__ __forceinline__ BxStream bx_stream(const BxProb& p, int rot, int& rot_out, int ou = 0, int okt = 0) {
    BxStream st;
    st.Mx = bx_mx(p); st.Kx = bx_kx(p);
    st.u = bx_units(p, st.Mx, st.Kx, ou, okt);
    const int xcd = blockIdx.x & 7, per = max(1, (int)gridDim.x >> 3);
    const int slot = (((int)blockIdx.x >> 3) - rot % per + per) % per;
    const int lo = (int)(((long)st.u.n * xcd) >> 3), hi = (int)(((long)st.u.n * (xcd + 1)) >> 3);
    st.first = lo + slot; st.step = per;
    st.count = hi - lo > slot ? (hi - lo - slot + per - 1) / per : 0;
    rot_out = (rot + (hi - lo)) % per;
    return st;
}

// up to two problems in one persistent launch (the dX / dW pair of a layer's backward): 4 compute waves + 2 loader waves per CU
template <int NP, int DBG = 0>
__global__ __launch_bounds__(64 * (BX_NC + BX_NL), 2) void bx3_kernel(BxProb p0, BxProb p1, int has1, int pair_policy) {
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int rot = 0, rot1 = 0;
    // the HEAVIER units first (longest-processing-time order inside a workgroup's run): the k-chunks of a dW before the dX tiles
    for (int pass = 0; pass < (has1 ? 2 : 1); ++pass) {
        const BxProb& p = (has1 && pass == 0) ? p1 : p0;
        int ou = 0, okt = 0;                           // the split-K problem of a pair sizes its chunks against the NT problem's units
        if (pair_policy && has1 && pass == 0 && p1.tn && !p0.tn) {
            const BxUnits u0 = bx_units(p0, bx_mx(p0), bx_kx(p0));
            ou = u0.n; okt = u0.kt_total;
        }
        const BxStream st = bx_stream(p, rot, rot1, ou, okt);
        rot = rot1;
        if (wave < BX_NC) {
            if (p.tn) bx_compute<true, NP, DBG>(p, st, wave); else bx_compute<false, NP, DBG>(p, st, wave);
        } else {
            if constexpr (DBG != 2) {
                if (p.tn) bx_loader<true, NP>(p, st, wave - BX_NC); else bx_loader<false, NP>(p, st, wave - BX_NC);
            } else {                                   // (probe: products only, on whatever the LDS holds)
                const int total = bx_total_tiles(p, st);
                for (int k = 0; k < total + 2; ++k) __builtin_amdgcn_s_barrier();
            }
        }
    }
}

// ---- operand producers -----------------------------------------------------------------------------------------------------------
// fp32 matrix [rows][ld] -> three bf16 planes (stand-alone conversion: C-ABI entry, tests, layer-level path; the model engine's
// producers write the planes in their own epilogues)
__global__ __launch_bounds__(256) void bx3_split_kernel(const float* __restrict__ x, int rows, const int* __restrict__ rows_dev, int ld,
                                                       uint16_t* __restrict__ pl, size_t pstride, int rows_cap, int np) {
    const int R = rows_dev ? min(*rows_dev, rows) : rows;
    const int ld4 = ld >> 2;
    const size_t n4 = (size_t)R * ld4;
    const BxOut o{pl, pstride, np, rows_cap};
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        const int r = (int)(i / ld4), c = (int)(i - (size_t)r * ld4) << 2;
        bx_store4(o, r, c, *reinterpret_cast<const float4*>(x + 4 * i));
    }
}

bool bx3_pair_policy() {
    static const bool v = [] { const char* e = getenv("EAGCN_BX3_PAIR_POLICY"); return !(e && e[0] == '0'); }();
    return v;
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

// ---- which of the two kernels (bx3.h) -------------------------------------------------------------------------------------------------
static int g_bx3_wide = [] { const char* e = getenv("EAGCN_BX3_WIDE"); return e ? atoi(e) : -1; }();      // -1 auto | 0 never | 1 always
static int bx3_wide_min(bool pair) {
    static const int v1 = [] { const char* e = getenv("EAGCN_BX3_WIDE_MIN"); return e ? atoi(e) : 80; }();
    static const int v2 = [] { const char* e = getenv("EAGCN_BX3_WIDE_MIN_PAIR"); return e ? atoi(e) : 50; }();
    return pair ? v2 : v1;
}
int bx3_pick_wide(const BxProb& p0, const BxProb* p1, int rows_hint) {
    if (g_bx3_wide >= 0) return g_bx3_wide ? 1 : 0;
    // Work of the launch in k-tiles of 128 x 128 tiles per CU.  The 256 x 128 kernel moves 25 % fewer operand bytes per flop into LDS
    // (both kernels run at the rate their CUs can fill LDS, about 33 GB/s each at these footprints: DESIGN.md section 4) and keeps two
    // compute waves per SIMD, but it has half as many tiles to hand out: it wins from about two waves of its tiles.  A launch with a
    // split-K product balances at chunk granularity and switches earlier.  Measured (profiles/r05_bx3w_*): 4809 rows (12 / 27 k-tiles
    // per CU forward / pair) 24 vs 37 us and 50 vs 61 us for the 128-tile kernel; 19 200 rows (46 / 108) 75 vs 70 and 167 vs 148 us for
    // the 256-tile kernel; 14 000 x 512 x 1040 (62 / 124) 90 vs 105 and 181 vs 163; HIV and C5 widths 1.2-1.35x for the 256-tile kernel.
    auto work = [&](const BxProb& p) -> double {
        int M = p.M, K = p.K;
        if (rows_hint > 0) { if (!p.tn) M = std::min(M, rows_hint); else K = std::min(K, rows_hint); }
        return (double)cdiv(M, BX3_BM) * cdiv(p.N, BX3_BN) * std::max(1, cdiv(K, BX_BK));
    };
    const double w = work(p0) + (p1 ? work(*p1) : 0.0);
    return w >= (double)bx3_wide_min(p1 != nullptr || p0.tn) * bx3_grid() ? 1 : 0;
}

int launch_bx3_split(const float* x, int rows, const int* rows_dev, int ld, uint16_t* planes, size_t pstride, int rows_cap, int np, hipStream_t s) {
    EAGCN_CHECK_ARG(x && planes && (ld & 3) == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0 && (reinterpret_cast<uintptr_t>(planes) & 15) == 0 &&
                        (pstride & 7) == 0, "bx3_split: operands must be 16-byte aligned with ld a multiple of 4");
    EAGCN_CHECK_ARG(rows <= rows_cap && pstride >= bx_plane_elems(rows_cap, ld), "bx3_split: %d rows do not fit the plane image (capacity %d rows)", rows, rows_cap);
    if (rows <= 0) return EAGCN_OK;
    const size_t n4 = (size_t)rows * ld / 4;
    bx3_split_kernel<<<(unsigned)std::max<size_t>(1, std::min<size_t>((n4 + 255) / 256, 4096)), 256, 0, s>>>(x, rows, rows_dev, ld, planes, pstride, rows_cap, np);
    EAGCN_LAUNCH_CHECK();
    return EAGCN_OK;
}

bool bx3_ok(const BxProb& p) {
    if (p.M <= 0 || p.N <= 0 || p.K <= 0 || !p.A.p || !p.B.p || !p.C) return false;
    if ((p.A.ld & 7) || (p.B.ld & 7) || (reinterpret_cast<uintptr_t>(p.A.p) & 15) || (reinterpret_cast<uintptr_t>(p.B.p) & 15)) return false;
    if ((p.A.pstride & 7) || (p.B.pstride & 7)) return false;
    if (p.A.pstride < bx_plane_elems(p.A.rows, p.A.ld) || p.B.pstride < bx_plane_elems(p.B.rows, p.B.ld)) return false;
    // the images hold the operands' rows; 32-bit byte offsets inside an image (buffer descriptor + voffset, tiles may reach 128 rows /
    // four panels beyond the extents)
    const int arows = p.tn ? p.K : p.M, brows = p.tn ? p.K : p.N;
    if (arows > p.A.rows || brows > p.B.rows) return false;
    if (((double)bx_plane_elems(p.A.rows, p.A.ld) + 5.0 * 64 * p.A.rows) * 2.0 >= 4.0e9 ||
        ((double)bx_plane_elems(p.B.rows, p.B.ld) + 5.0 * 64 * p.B.rows) * 2.0 >= 4.0e9) return false;
    if (!p.tn) return (p.K & 7) == 0 && p.K <= p.A.ld && p.K <= p.B.ld && !p.K_dev;
    return !p.M_dev && p.M <= p.A.ld && p.N <= p.B.ld && p.splits >= 1 && (p.splits == 1 || p.slab >= (size_t)p.M * p.ldc);
}

template <int NP, int DBG = 0>
static int bx3_launch_cfg(const BxProb& p0, const BxProb* p1, hipStream_t s) {
    constexpr int lds = BX_NS * BX_STAGE;
    static bool attr_done = false;
    if (!attr_done) {
        EAGCN_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&bx3_kernel<NP, DBG>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        attr_done = true;
    }
    bx3_kernel<NP, DBG><<<bx3_grid(), 64 * (BX_NC + BX_NL), lds, s>>>(p0, p1 ? *p1 : p0, p1 ? 1 : 0, bx3_pair_policy() ? 1 : 0);
    EAGCN_LAUNCH_CHECK();
    return EAGCN_OK;
}

// np = 3: fp32-equivalent products (three planes); np = 1: plain bf16 operands (one plane, one product)
int launch_bx3(const BxProb& p0, const BxProb* p1, int np, hipStream_t s, double work, int prof_tag, int wide) {
    EAGCN_CHECK_ARG(bx3_ok(p0) && (!p1 || bx3_ok(*p1)), "bx3 gemm: operands not aligned / extents unsupported");
    EAGCN_CHECK_ARG(np == 1 || np == 3, "bx3 gemm: 1 or 3 planes");
    ProfScope ps(prof_tag, s, work);
    if (wide) return launch_bx3w(p0, p1, np, s);
    static const int dbg = [] { const char* e = getenv("EAGCN_BX3_DBG"); return e ? atoi(e) : 0; }();     // probes (wrong results!)
    if (np == 3 && dbg == 1) return bx3_launch_cfg<3, 1>(p0, p1, s);
    if (np == 3 && dbg == 2) return bx3_launch_cfg<3, 2>(p0, p1, s);
    return np == 3 ? bx3_launch_cfg<3>(p0, p1, s) : bx3_launch_cfg<1>(p0, p1, s);
}

}  // namespace eagcn

using namespace eagcn;

/* plane images (PANEL-MAJOR, include/eagcn_hip.h) of a row-major fp32 matrix [rows][ld]: three bf16 planes (np = 3: x = x0 + x1 + x2
 * exactly) or one (np = 1: round to nearest even), plane q at planes + q * plane_stride (elements), images of row capacity rows_cap */
extern "C" int eagcn_bx3_split(const float* x, int rows, int ld, uint16_t* planes, size_t plane_stride, int rows_cap, int np, void* stream) {
    EAGCN_CHECK_ARG(np == 1 || np == 3, "eagcn_bx3_split: np must be 1 or 3");
    return launch_bx3_split(x, rows, nullptr, ld, planes, plane_stride, rows_cap, np, (hipStream_t)stream);
}

extern "C" size_t eagcn_bx3_plane_elems(int rows_cap, int ld) { return bx_plane_elems(rows_cap, ld); }

/* C = op(A) . op(B) from bf16 plane images.  tn = 0: C[M,N] = A[M,K] . B[N,K]^T (images of [a_rows >= M][lda >= K], [b_rows >= N][ldb >= K]);
 * tn = 1: C[M,N] = A[K,M]^T . B[K,N] (images of [a_rows >= K][lda >= M], [b_rows >= K][ldb >= N]) written as partial slabs
 * C + z * slab (their sum is the product; eagcn_bx3_used_splits of them are written). */
extern "C" int eagcn_gemm_bx3(int tn, int M, int N, int K, const uint16_t* A, size_t a_pstride, int lda, int a_rows, const uint16_t* B,
                              size_t b_pstride, int ldb, int b_rows, float* C, int ldc, int splits, size_t slab, int np, void* stream) {
    BxProb p;
    memset(&p, 0, sizeof(p));
    p.A = BxPlanes{A, a_pstride, lda, a_rows}; p.B = BxPlanes{B, b_pstride, ldb, b_rows}; p.C = C; p.ldc = ldc;
    p.M = M; p.N = N; p.K = K; p.tn = tn; p.splits = tn ? std::max(1, splits) : 1; p.slab = slab;
    return launch_bx3(p, nullptr, np, (hipStream_t)stream, 2.0 * M * N * K, PROF_GEMM, bx3_pick_wide(p, nullptr, 0));
}

/* -1: the library picks the kernel of a plane-GEMM launch by its size (default) | 0: always the 128 x 128 kernel | 1: always the
 * 256 x 128 kernel (csrc/gemm_bx3w.hip).  Returns the previous setting. */
extern "C" int eagcn_set_bx3_wide(int mode) {
    const int old = g_bx3_wide;
    g_bx3_wide = mode < 0 ? -1 : (mode ? 1 : 0);
    return old;
}

static BxProb bx3_dims(int tn, int M, int N, int K) {
    BxProb p;
    memset(&p, 0, sizeof(p));
    p.M = M; p.N = N; p.K = K; p.tn = tn;
    return p;
}
extern "C" int eagcn_bx3_used_splits(int splits, int M, int N, int K) {
    const int wide = bx3_pick_wide(bx3_dims(1, M, N, K), nullptr, 0);
    return bx3_used_splits(splits < 1 ? 1 : splits, K, cdiv(M, wide ? BX3W_BM : BX3_BM) * cdiv(N, wide ? BX3W_BN : BX3_BN));
}
/* the same for the TN problem (M, N, K) of eagcn_gemm_bx3_pair, whose chunks are sized against the NT problem (M0, N0, K0) */
extern "C" int eagcn_bx3_pair_used_splits(int splits, int M, int N, int K, int M0, int N0, int K0) {
    const BxProb q = bx3_dims(1, M, N, K);
    const int wide = bx3_pick_wide(bx3_dims(0, M0, N0, K0), &q, 0);
    const int bm = wide ? BX3W_BM : BX3_BM, bn = wide ? BX3W_BN : BX3_BN;
    return bx3_used_splits(splits < 1 ? 1 : splits, K, cdiv(M, bm) * cdiv(N, bn), bx3_pair_policy() ? cdiv(M0, bm) * cdiv(N0, bn) : 0,
                           std::max(1, cdiv(K0, BX_BK)), std::max(1, bx3_grid() >> 3));
}

/* the dX / dW pair of a layer's backward in ONE persistent launch: problem 0 NT, problem 1 TN with split-K slabs */
extern "C" int eagcn_gemm_bx3_pair(int M0, int N0, int K0, const uint16_t* A0, size_t a0_pstride, int lda0, int a0_rows, const uint16_t* B0,
                                   size_t b0_pstride, int ldb0, int b0_rows, float* C0, int ldc0, int M1, int N1, int K1, const uint16_t* A1,
                                   size_t a1_pstride, int lda1, int a1_rows, const uint16_t* B1, size_t b1_pstride, int ldb1, int b1_rows,
                                   float* C1, int ldc1, int splits, size_t slab, int np, void* stream) {
    BxProb p, q;
    memset(&p, 0, sizeof(p));
    memset(&q, 0, sizeof(q));
    p.A = BxPlanes{A0, a0_pstride, lda0, a0_rows}; p.B = BxPlanes{B0, b0_pstride, ldb0, b0_rows}; p.C = C0; p.ldc = ldc0;
    p.M = M0; p.N = N0; p.K = K0; p.tn = 0; p.splits = 1;
    q.A = BxPlanes{A1, a1_pstride, lda1, a1_rows}; q.B = BxPlanes{B1, b1_pstride, ldb1, b1_rows}; q.C = C1; q.ldc = ldc1;
    q.M = M1; q.N = N1; q.K = K1; q.tn = 1; q.splits = std::max(1, splits); q.slab = slab;
    return launch_bx3(p, &q, np, (hipStream_t)stream, 2.0 * M0 * N0 * K0 + 2.0 * M1 * N1 * K1, PROF_GEMM_PAIR, bx3_pick_wide(p, &q, 0));
}
