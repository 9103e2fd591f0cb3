// The plane GEMM of csrc/gemm_bx3.hip for launches with SEVERAL waves of tiles: 256 x 128 workgroup tile, EIGHT compute waves -- two
// per SIMD, staggered by half a k-tile -- and no loader waves.
//
// Why (profiles/r04_bx3_sq.txt): the 128 x 128 kernel keeps ONE compute wave per SIMD; per k-tile that
// wave has 0.73 us of MFMA issue and measured 1.39 us, the SIMD's matrix pipe idling whenever its only wave waits -- for LDS
// fragments, at the two barriers per k-tile, in the epilogue store.  Here every SIMD holds TWO compute waves that run the same
// stream of half-steps  R(h): read the 12 operand fragments of k16-step h from LDS;  M(h): 24 MFMAs on them  -- group G1 (waves
// 4-7, the partners of waves 0-3 on their SIMDs: MI355X_MICROARCH.md "Two waves per SIMD") half a k-tile behind group G0:
//
//     G0:  R(k,0) M(k,0) R(k,1) M(k,1) | B_k+1 |  R(k+1,0) M(k+1,0) ...
//     G1:         R(k,0) M(k,0) R(k,1) | B_k+1 |  M(k,1)   R(k+1,0) M(k+1,0) ...
//
// The ONE barrier per k-tile cuts G1's half-step between its reads and its MFMAs: behind the barrier G1 multiplies from registers
// at once while G0 issues the LDS-DMA of the tile after next and reads its first fragments -- the matrix pipe has work across the
// barrier, and after it the two waves of a SIMD alternate by themselves (one waits for LDS while the other multiplies).  A wave
// holds ONE fragment set (48 registers; the 128 x 128 kernel double-buffers 96 inside its only wave) next to the 128 accumulator
// registers (`hi` / `lo` pair, gemm_bx3.hip), which is what lets two of them share a SIMD's 512 registers.
// LDS: 2 stages x 72 KB (A 256 rows + B 128 rows, 64 B per row and plane, three planes).  Tile k+2 is requested right behind
// barrier B_k+1 into the stage tile k was read from (G0 finished it before the barrier, G1's last reads of it are waited for in
// front of the barrier) and has a whole k-tile to land before every wave's `vmcnt(0)` in front of B_k+2.  Each wave requests nine
// 1 KB pieces per k-tile (per plane: A pieces w and w + 8, B piece w).
// Everything else is gemm_bx3.hip's: the panel-major plane images (bx3.h) ARE the LDS image; NT / TN forms, fragment reads, the two
// accumulators, k beyond K requested out of range, the persistent XCD-contiguous unit lists and the split-K chunk policy (in units
// of THIS kernel's tiles: the consumers of the slabs are told the tile shape, kernels.h).
#include <stdlib.h>

#include <algorithm>

#include "bx3.h"
#include "common.h"
#include "kernels.h"

namespace eagcn {

typedef __bf16 bw_bf16x8 __attribute__((ext_vector_type(8)));
typedef float bw_f32x16 __attribute__((ext_vector_type(16)));
typedef uint32_t bw_u32x4 __attribute__((ext_vector_type(4)));
typedef short bw_s16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* bw_lds_ptr;

constexpr int BW_BK = 32;
constexpr int BW_BM = BX3W_BM, BW_BN = BX3W_BN;                  // 256 x 128 (bx3.h)
constexpr int BW_NW = 8;                                         // compute waves: 4 (rows) x 2 (columns = the group), 64 x 64 each
constexpr int BW_APL = BW_BM * 64, BW_BPL = BW_BN * 64;          // bytes of one plane image of a k-tile: 16 KB / 8 KB
constexpr int BW_STAGE = 3 * (BW_APL + BW_BPL);                  // 72 KB
constexpr int BW_NS = 2;                                         // 144 KB
constexpr int BX3W_VAR_DEFAULT = 4;                                // DMAPOS 2, no priority flips (measured best of the five: profiles/r05_bx3w_variants.txt)
static_assert(BW_APL / 1024 == 2 * BW_NW && BW_BPL / 1024 == BW_NW, "piece hand-out: two A pieces and one B piece per wave and plane");

extern __shared__ __attribute__((aligned(1024))) unsigned char bw_smem[];

// buffer descriptor over `bytes` bytes from `base` (raw buffer: offsets are range-checked, out-of-range lanes read zero)
__device__ __forceinline__ bw_u32x4 bw_rsrc(const void* base, unsigned bytes) {
    const uint64_t a = (uint64_t)(size_t)base;
    bw_u32x4 r;
    r[0] = __builtin_amdgcn_readfirstlane((unsigned)a);
    r[1] = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32) & 0xFFFFu);
    r[2] = __builtin_amdgcn_readfirstlane(bytes);
    r[3] = 0x00020000u;
    return r;
}
// one LDS-DMA request: 64 lanes x 16 bytes from descriptor + voff (per lane) to LDS bytes [lds_dst, lds_dst + 1024), lane-linear.
// M0 carries the LDS address; it is the compiler's register: saved and restored inside the statement (cdna_hip_programming.md 5.7)
__device__ __forceinline__ void bw_dma16(const bw_u32x4 rsrc, unsigned voff, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(rsrc), "s"(__builtin_amdgcn_readfirstlane(lds_dst)) : "memory");
}

// ---- work units (gemm_bx3.hip, with this kernel's tile) --------------------------------------------------------------------------------
struct BwUnits { int tiles_m, tiles_n, splits, kt_total, kt_per; int n; };
__device__ __forceinline__ int bw_mx(const BxProb& p) { return (!p.tn && p.M_dev) ? min(*p.M_dev, p.M) : p.M; }
__device__ __forceinline__ int bw_kx(const BxProb& p) { return (p.tn && p.K_dev) ? min(*p.K_dev, p.K) : p.K; }
__device__ __forceinline__ BwUnits bw_units(const BxProb& p, int Mx, int Kx, int ou = 0, int okt = 0) {
    BwUnits u;
    u.tiles_m = (Mx + BW_BM - 1) / BW_BM;
    u.tiles_n = (p.N + BW_BN - 1) / BW_BN;
    u.kt_total = max(1, (Kx + BW_BK - 1) / BW_BK);
    u.splits = p.tn ? bx3_used_splits_wave(max(1, p.splits), Kx, u.tiles_m * u.tiles_n, ou, okt, max(1, (int)gridDim.x >> 3)) : 1;
    u.kt_per = (u.kt_total + u.splits - 1) / u.splits;
    u.n = u.tiles_m * u.tiles_n * u.splits;
    return u;
}
struct BwStream { BwUnits u; int Mx, Kx, first, step, count; };
struct BwUnit { int tm, tn, kt0, nk, z; };
__host__ __device__ __forceinline__ BwUnit bw_unit(const BxProb& p, const BwStream& st, int j) {
    const int id = st.first + j * st.step;
    BwUnit r;
    if (!p.tn) {
        r.tm = id / st.u.tiles_n; r.tn = id - r.tm * st.u.tiles_n; r.kt0 = 0; r.nk = st.u.kt_total; r.z = 0;
    } else {                                           // k-chunk major (the tiles of a chunk share its rows in their XCD's L2)
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
__device__ __forceinline__ int bw_total_tiles(const BxProb& p, const BwStream& st) {
    if (!p.tn) return st.count * st.u.kt_total;
    int t = 0;
    for (int j = 0; j < st.count; ++j) t += bw_unit(p, st, j).nk;
    return t;
}
__device__ __forceinline__ BwStream bw_stream(const BxProb& p, int rot, int& rot_out, int ou = 0, int okt = 0) {
    BwStream st;
    st.Mx = bw_mx(p); st.Kx = bw_kx(p);
    st.u = bw_units(p, st.Mx, st.Kx, ou, okt);
    const int xcd = blockIdx.x & 7, per = max(1, (int)gridDim.x >> 3);
    const int slot = (((int)blockIdx.x >> 3) - rot % per + per) % per;
    const int lo = (int)(((long)st.u.n * xcd) >> 3), hi = (int)(((long)st.u.n * (xcd + 1)) >> 3);
    st.first = lo + slot; st.step = per;
    st.count = hi - lo > slot ? (hi - lo - slot + per - 1) / per : 0;
    rot_out = (rot + (hi - lo)) % per;
    return st;
}

// ---- one compute wave --------------------------------------------------------------------------------------------------------------------
// wr = row block (0..3) of the 4 x 2 wave grid, G = column block = group (compile time: the two groups run differently rotated loops)
template <bool TN, int NP, int G, int VAR, int DBG>
__device__ __forceinline__ void bw_wave(const BxProb& p, const BwStream& st, const int wr) {
    constexpr int PRIO = VAR & 1, DMAPOS = VAR >> 1;   // (A/B switches: EAGCN_BX3W_VAR)
    const int lane = threadIdx.x & 63;
    const int w8 = G * 4 + wr;                         // 0..7: which DMA pieces this wave requests
    // ---- LDS-DMA side: buffer descriptors (one per operand plane over the whole image), per-lane source offsets
    const unsigned apanel = (unsigned)p.A.rows * 64u, bpanel = (unsigned)p.B.rows * 64u;      // bytes of one 32-column panel
    const unsigned abytes = (unsigned)((p.A.ld + 31) >> 5) * apanel, bbytes = (unsigned)((p.B.ld + 31) >> 5) * bpanel;
    // The requests are issued through inline asm (bw_dma16): hipcc orders every LDS read behind ANY LDS-DMA it has seen this wave
    // issue (a `vmcnt(0)` in front of the first fragment read behind the requests of the tile after next -- found in the ISA of the
    // builtin form of this loop), and this pipeline lives on requests that stay in flight across reads and barriers.  The waits
    // that order them are the explicit ones below: own `vmcnt`, then the barrier, then the reads.
    bw_u32x4 ra[3], rb[3];
#pragma unroll
    for (int q = 0; q < NP; ++q) {
        ra[q] = bw_rsrc(p.A.p + (size_t)q * p.A.pstride, abytes);
        rb[q] = bw_rsrc(p.B.p + (size_t)q * p.B.pstride, bbytes);
    }
    // piece pc of an operand's tile image = 1 KB of consecutive source bytes, lane-linear (NT: rows 16 pc .. of the k-tile's panel;
    // TN: k-rows 16 (pc & 1) .. of the tile's panel pc >> 1); this wave: A pieces w8 and w8 + 8, B piece w8
    unsigned rel_a0, rel_a1, rel_b;
    if constexpr (!TN) {
        rel_a0 = (unsigned)(w8 * 1024 + lane * 16); rel_a1 = rel_a0 + 8u * 1024u; rel_b = rel_a0;
    } else {
        rel_a0 = (unsigned)(w8 >> 1) * apanel + (unsigned)((w8 & 1) * 1024 + lane * 16);
        rel_a1 = rel_a0 + 4u * apanel;
        rel_b = (unsigned)(w8 >> 1) * bpanel + (unsigned)((w8 & 1) * 1024 + lane * 16);
    }
    const unsigned astep = TN ? BW_BK * 64u : apanel, bstep = TN ? BW_BK * 64u : bpanel;      // k-tile to k-tile
    const unsigned OOB = 0xFFFFFF00u;                  // beyond every num_records (bx3_ok keeps the images below 4e9 bytes)
    // k offset of this lane's 16 bytes inside a k-tile (the same for all three of the wave's pieces): k beyond the actual K is requested
    // out of range and arrives as zeros.  NT: the lane's LOGICAL chunk (the panel image is XOR-swizzled by (row >> 2) & 3 = (lane >> 4) & 3
    // for every piece); TN: the lane's k-row
    const int koff = TN ? (w8 & 1) * 16 + (lane >> 2) : 8 * ((lane & 3) ^ ((lane >> 4) & 3));
    const int total = bw_total_tiles(p, st);
    const unsigned lds0 = (unsigned)(size_t)(bw_lds_ptr)bw_smem;       // LDS byte address of the stages
    int jd = -1, leftd = 0, kposd = 0;
    unsigned oa = 0, ob = 0;
    // a tile's nine requests: dma_prepare() advances the iterator and forms the three source offsets, dma_piece(i) issues request i
    // (i = 3 q + {A piece w8, A piece w8 + 8, B piece w8}) -- separately, so that the requests can be placed between MFMAs
    bool dlive = false;
    unsigned dva0 = 0, dva1 = 0, dvb = 0, dsb = 0;
    auto dma_prepare = [&](int stage) __attribute__((always_inline)) {
        while (leftd == 0 && jd + 1 < st.count) {      // next unit that has k-tiles
            ++jd;
            const BwUnit un = bw_unit(p, st, jd);
            leftd = un.nk;
            kposd = un.kt0 * BW_BK;
            if constexpr (!TN) {                       // panel kt0, rows of the tile
                oa = (unsigned)un.kt0 * apanel + (unsigned)(un.tm * BW_BM) * 64u;
                ob = (unsigned)un.kt0 * bpanel + (unsigned)(un.tn * BW_BN) * 64u;
            } else {                                   // the tile's first panel, k-row kt0 * 32
                oa = (unsigned)(un.tm * (BW_BM / 32)) * apanel + (unsigned)(un.kt0 * BW_BK) * 64u;
                ob = (unsigned)(un.tn * (BW_BN / 32)) * bpanel + (unsigned)(un.kt0 * BW_BK) * 64u;
            }
        }
        dlive = leftd != 0;
        if (!dlive) return;
        const bool ok = kposd + koff < st.Kx;
        dsb = lds0 + (unsigned)(stage * BW_STAGE + w8 * 1024);
        dva0 = ok ? oa + rel_a0 : OOB; dva1 = ok ? oa + rel_a1 : OOB; dvb = ok ? ob + rel_b : OOB;
        oa += astep; ob += bstep; kposd += BW_BK; --leftd;
        if constexpr (DBG == 1) dlive = false;         // (probe: no requests)
    };
    auto dma_piece = [&](const int i) __attribute__((always_inline)) {
        const int q = i / 3, r = i - 3 * q;
        if (r == 0) bw_dma16(ra[q], dva0, dsb + q * BW_APL);
        else if (r == 1) bw_dma16(ra[q], dva1, dsb + q * BW_APL + 8 * 1024);
        else bw_dma16(rb[q], dvb, dsb + 3 * BW_APL + q * BW_BPL);
    };
    auto dma_all = [&]() __attribute__((always_inline)) {
        if (!dlive) return;
#pragma unroll
        for (int i = 0; i < 3 * NP; ++i) dma_piece(i);
        dlive = false;
    };
    // ---- fragment read offsets (bytes inside a plane image), as in gemm_bx3.hip with a 4 x 2 wave grid
    int fa[2], fb[2], fah[2], fbh[2];                  // NT: per k16-step s; TN: per 32-column tile t (lo / hi: k-rows +0 / +4)
    if constexpr (!TN) {
        const int i = lane & 31, kg = lane >> 5, x = (i >> 2) & 3;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            fa[s] = (wr * 64 + i) * 64 + (((2 * s + kg) ^ x) << 4);
            fb[s] = (G * 64 + i) * 64 + (((2 * s + kg) ^ x) << 4);
        }
    } else {
        const int b = lane >> 4, c = lane & 15;
        const int row = 8 * (b >> 1) + (c >> 2), ch = 2 * (b & 1) + ((c & 3) >> 1), half = 8 * (c & 1);
        const int lo_off = row * 64 + ((ch ^ (2 * (b >> 1))) << 4) + half;
        const int hi_off = (row + 4) * 64 + ((ch ^ (2 * (b >> 1) + 1)) << 4) + half;
        // (tile t of a wave = panel 2 w + t: + t * 2048 bytes, a compile-time offset of the read -- only FOUR offset registers live
        //  across the k-loop; eight put the 128 x 128 kernel two registers over its 256 and it re-loaded them from scratch every k-tile)
        fa[0] = (2 * wr) * 2048 + lo_off; fah[0] = (2 * wr) * 2048 + hi_off;
        fb[0] = (2 * G) * 2048 + lo_off; fbh[0] = (2 * G) * 2048 + hi_off;
        fa[1] = fb[1] = fah[1] = fbh[1] = 0;
    }
    auto frag = [&](const unsigned char* plane, int t, int s, bool is_a) __attribute__((always_inline)) -> bw_bf16x8 {
        if constexpr (!TN) {
            const int off = (is_a ? fa[s] : fb[s]) + t * 32 * 64;
            return __builtin_bit_cast(bw_bf16x8, *reinterpret_cast<const bw_u32x4*>(plane + off));
        } else {
            const int off = (is_a ? fa[0] : fb[0]) + t * 2048 + 16 * s * 64, offh = (is_a ? fah[0] : fbh[0]) + t * 2048 + 16 * s * 64;
            const bw_s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((bw_s16x4 __attribute__((address_space(3)))*)(plane + off));
            const bw_s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((bw_s16x4 __attribute__((address_space(3)))*)(plane + offh));
            typedef short s16x8 __attribute__((ext_vector_type(8)));
            const s16x8 v = (s16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            return __builtin_bit_cast(bw_bf16x8, v);
        }
    };
    bw_f32x16 acc[2][2], lo[2][2];                     // (two accumulators per output tile: gemm_bx3.hip on the bf16 MFMA's truncating accumulate)
    auto zero_acc = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int jj = 0; jj < 2; ++jj)
#pragma unroll
                for (int r = 0; r < 16; ++r) { acc[i][jj][r] = 0.f; lo[i][jj][r] = 0.f; }
    };
    bw_bf16x8 af[2][NP], bf[2][NP];
    // reads in the order the products below consume them: (a2, b0), (a1, b1), (a0, b2)
    auto load_set = [&](int stg, int s) __attribute__((always_inline)) {
        if constexpr (DBG == 3) { if (stg >= 0) return; stg = 0; }      // (probe: no fragment reads inside the loop)
        const unsigned char* sa = bw_smem + stg * BW_STAGE;
        const unsigned char* sbp = sa + 3 * BW_APL;
#pragma unroll
        for (int u = 0; u < NP; ++u) {
            const int qa = NP - 1 - u, qb = u;
#pragma unroll
            for (int t = 0; t < 2; ++t) af[t][qa] = frag(sa + qa * BW_APL, t, s, true);
#pragma unroll
            for (int t = 0; t < 2; ++t) bf[t][qb] = frag(sbp + qb * BW_BPL, t, s, false);
        }
    };
    // the 24 MFMAs of a half-step; with_dma: the pending tile's requests between the product groups, two at a time (an LDS-DMA request
    // holds its wave's issue for 60-180 cycles, MI355X_MICROARCH.md: nine in a row right behind the barrier kept BOTH waves of every SIMD
    // out of the matrix pipe -- 568 us with, 387 us without the requests at 100 000 x 512 x 1024, profiles/r05_bx3w_ablation.txt)
    auto mma_set = [&](const bool with_dma) __attribute__((always_inline)) {
#define EAGCN_BW_PROD(ACC, PA, PB)                                                                                     \
    _Pragma("unroll") for (int i = 0; i < 2; ++i) _Pragma("unroll") for (int jj = 0; jj < 2; ++jj)                     \
        ACC[i][jj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][PA], bf[jj][PB], ACC[i][jj], 0, 0, 0);
#define EAGCN_BW_DMA2(I0, I1)                                                                                          \
    if (with_dma && dlive) {                                                                                           \
        __builtin_amdgcn_sched_barrier(0);                                                                             \
        if ((I0) < 3 * NP) dma_piece(I0);                                                                              \
        if ((I1) < 3 * NP) dma_piece(I1);                                                                              \
        __builtin_amdgcn_sched_barrier(0);                                                                             \
    }
        if constexpr (DBG == 2) {                      // (probe: no products; the fragments stay live)
#pragma unroll
            for (int q = 0; q < NP; ++q)
#pragma unroll
                for (int t = 0; t < 2; ++t) { asm volatile("" ::"v"(af[t][q])); asm volatile("" ::"v"(bf[t][q])); }
            if (with_dma) dma_all();
            return;
        }
        if constexpr (NP == 3) {
            EAGCN_BW_PROD(lo, 2, 0)
            EAGCN_BW_DMA2(0, 1)
            EAGCN_BW_PROD(lo, 1, 1)
            EAGCN_BW_DMA2(2, 3)
            EAGCN_BW_PROD(lo, 0, 2)
            EAGCN_BW_DMA2(4, 5)
            EAGCN_BW_PROD(lo, 1, 0)
            EAGCN_BW_DMA2(6, 7)
            EAGCN_BW_PROD(lo, 0, 1)
            EAGCN_BW_DMA2(8, 99)
            EAGCN_BW_PROD(acc, 0, 0)
        } else {
            EAGCN_BW_PROD(acc, 0, 0)
            if (with_dma) dma_all();
        }
        if (with_dma) dlive = false;
#undef EAGCN_BW_PROD
#undef EAGCN_BW_DMA2
    };
    // D layout: col = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)
    auto store_unit = [&](const BwUnit& un) __attribute__((always_inline)) {
        float* __restrict__ Cz = p.C + (size_t)un.z * p.slab;
        const int Mlim = TN ? p.M : st.Mx;
        const int m0 = un.tm * BW_BM + wr * 64, n0 = un.tn * BW_BN + G * 64;
        if (m0 + 64 <= Mlim && n0 + 64 <= p.N) {       // whole block inside the matrix: no predicates
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int jj = 0; jj < 2; ++jj) {
                    float* cp = Cz + (size_t)(m0 + i * 32 + 4 * (lane >> 5)) * p.ldc + n0 + jj * 32 + (lane & 31);
#pragma unroll
                    for (int r = 0; r < 16; ++r) cp[(size_t)((r & 3) + 8 * (r >> 2)) * p.ldc] = acc[i][jj][r] + lo[i][jj][r];
                }
        } else if (m0 < Mlim && n0 < p.N) {
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
    // ---- the stream -------------------------------------------------------------------------------------------------------------------
    zero_acc();
    int j = 0;
    BwUnit un = st.count > 0 ? bw_unit(p, st, 0) : BwUnit{0, 0, 0, 1 << 30, 0};
    while (j < st.count && un.nk == 0) {               // (empty k-chunks store zeros)
        store_unit(un);
        if (++j < st.count) un = bw_unit(p, st, j);
    }
    int rem = un.nk;
    if (total == 0) return;                            // (uniform over the workgroup: no barrier is missed)
    dma_prepare(0); dma_all();
    if (total > 1) {
        dma_prepare(1); dma_all();
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * NP) : "memory");      // tile 0 has landed (this wave's pieces of it)
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();                      // B_0: ... and everybody else's
    __builtin_amdgcn_sched_barrier(0);
    // barrier B_k+1: tile k + 1 has landed (every wave waited for its own pieces), nobody reads tile k's stage any more -> tile k + 2 may
    // be requested.  WHERE the nine requests go (DMAPOS):
    //   0  all nine right behind the barrier, both groups (first version)
    //   1  G1: between the MFMAs of M(k,1), which it runs right behind the barrier; G0: as a block behind the reads R(k+1,0) (the reads'
    //      latency covers part of the requests' issue time; G1 feeds the matrix pipe meanwhile)
    //   2  G1 as 1; G0: between the MFMAs of M(k+1,0)
    auto sync_point = [&](int k) __attribute__((always_inline)) {
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        if (k + 2 < total) {
            dma_prepare(k & 1);
            if constexpr (DMAPOS == 0) dma_all();
        }
    };
    auto unit_end = [&]() __attribute__((always_inline)) {
        store_unit(un);
        zero_acc();
        do {
            if (++j < st.count) un = bw_unit(p, st, j); else un.nk = 1 << 30;
            if (j < st.count && un.nk == 0) store_unit(un);
        } while (j < st.count && un.nk == 0);
        rem = un.nk;
    };
    if constexpr (DBG == 3) load_set(-1, 0);
    for (int k = 0; k < total; ++k) {
        const int stage = k & 1;
        load_set(stage, 0);
        if constexpr (G == 0 && DMAPOS == 1) { __builtin_amdgcn_sched_barrier(0); dma_all(); __builtin_amdgcn_sched_barrier(0); }
        if constexpr (PRIO) __builtin_amdgcn_s_setprio(1);
        mma_set(G == 0 && DMAPOS == 2);
        if constexpr (PRIO) __builtin_amdgcn_s_setprio(0);
        load_set(stage, 1);
        if constexpr (G == 1) sync_point(k);
        if constexpr (PRIO) __builtin_amdgcn_s_setprio(1);
        mma_set(G == 1 && DMAPOS != 0);
        if constexpr (PRIO) __builtin_amdgcn_s_setprio(0);
        if constexpr (G == 0) sync_point(k);
        if (--rem == 0) unit_end();
    }
}

// up to two problems in one persistent launch (the dX / dW pair of a layer's backward), as bx3_kernel
template <int NP, int VAR, int DBG>
__global__ __launch_bounds__(64 * BW_NW, 2) void bx3w_kernel(BxProb p0, BxProb p1, int has1, int pair_policy) {
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int rot = 0, rot1 = 0;
    for (int pass = 0; pass < (has1 ? 2 : 1); ++pass) {
        const BxProb& p = (has1 && pass == 0) ? p1 : p0;
        int ou = 0, okt = 0;                           // the split-K problem of a pair sizes its chunks against the NT problem's units
        if (pair_policy && has1 && pass == 0 && p1.tn && !p0.tn) {
            const BwUnits u0 = bw_units(p0, bw_mx(p0), bw_kx(p0));
            ou = u0.n; okt = u0.kt_total;
        }
        const BwStream st = bw_stream(p, rot, rot1, ou, okt);
        rot = rot1;
        if (wave < 4) {
            if (p.tn) bw_wave<true, NP, 0, VAR, DBG>(p, st, wave); else bw_wave<false, NP, 0, VAR, DBG>(p, st, wave);
        } else {
            if (p.tn) bw_wave<true, NP, 1, VAR, DBG>(p, st, wave - 4); else bw_wave<false, NP, 1, VAR, DBG>(p, st, wave - 4);
        }
        // (no barrier between the passes: behind the last barrier of a pass no wave reads LDS any more -- sync_point waits for a wave's
        //  reads in front of every barrier -- so the next pass's first requests may overwrite both stages at once)
    }
}

template <int NP, int VAR, int DBG = 0>
static int bx3w_launch_cfg(const BxProb& p0, const BxProb* p1, hipStream_t s) {
    constexpr int lds = BW_NS * BW_STAGE;
    static bool attr_done = false;
    if (!attr_done) {
        EAGCN_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&bx3w_kernel<NP, VAR, DBG>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        attr_done = true;
    }
    bx3w_kernel<NP, VAR, DBG><<<bx3_grid(), 64 * BW_NW, lds, s>>>(p0, p1 ? *p1 : p0, p1 ? 1 : 0, bx3_pair_policy() ? 1 : 0);
    EAGCN_LAUNCH_CHECK();
    return EAGCN_OK;
}

int launch_bx3w(const BxProb& p0, const BxProb* p1, int np, hipStream_t s) {
    static const int var = [] { const char* e = getenv("EAGCN_BX3W_VAR"); return e ? atoi(e) : BX3W_VAR_DEFAULT; }();      // A/B switch: bit 0 s_setprio around the MFMA blocks, bits 1.. DMAPOS
    static const int dbg = [] { const char* e = getenv("EAGCN_BX3W_DBG"); return e ? atoi(e) : 0; }();         // probes (wrong results!)
    if (np == 3 && dbg == 1) return bx3w_launch_cfg<3, BX3W_VAR_DEFAULT, 1>(p0, p1, s);
    if (np == 3 && dbg == 2) return bx3w_launch_cfg<3, BX3W_VAR_DEFAULT, 2>(p0, p1, s);
    if (np == 3 && dbg == 3) return bx3w_launch_cfg<3, BX3W_VAR_DEFAULT, 3>(p0, p1, s);
    if (np == 3) {
        switch (var) {
            case 0: return bx3w_launch_cfg<3, 0>(p0, p1, s);
            case 1: return bx3w_launch_cfg<3, 1>(p0, p1, s);
            case 2: return bx3w_launch_cfg<3, 2>(p0, p1, s);
            case 3: return bx3w_launch_cfg<3, 3>(p0, p1, s);
            case 4: return bx3w_launch_cfg<3, 4>(p0, p1, s);
            default: return bx3w_launch_cfg<3, 5>(p0, p1, s);
        }
    }
    return bx3w_launch_cfg<1, BX3W_VAR_DEFAULT>(p0, p1, s);
}

}  // namespace eagcn
