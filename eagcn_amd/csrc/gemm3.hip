// Wave-autonomous, balanced ("stream-K") fp32 GEMM on the CDNA4 matrix cores for the layer products.
//
// Why not the classic LDS-tiled workgroup kernel (gemm.hip): v_mfma_f32_16x16x4_f32 is exact fp32 at the fp32 VECTOR
// rate (32 cycles per instruction and SIMD), 16x slower than the bf16 forms, so operand delivery is cheap relative to the
// matrix pipe -- what costs is everything that PARKS a wave.  Counters of the workgroup kernels (profiles/r02_sq_counters.txt):
// waves spend 27-45 % of their life in s_waitcnt / s_barrier per 16 MFMAs (LDS round trips + the barrier that couples four
// waves on four SIMDs, each fighting other workgroups' waves for its matrix pipe) and the pipes are only 50-56 % busy.
// Here every WAVE owns a 64x64 output tile (16 accumulators = 64 VGPRs), takes its operand fragments STRAIGHT from
// global memory into the registers the MFMAs read (no LDS, no barrier, nothing shared between waves), 64 MFMAs per
// 16-wide k-step behind one prefetched register set:
//   NT form  C[m,n] = sum_k A[m,k] B[n,k]  (both K-contiguous: the forward transform with the pre-transposed weight,
//            and dX = dP.W^T):  lane (li, q) loads the float4 A[m0 + 16 i + li][k0 + 4q .. +3]; MFMA step s multiplies the
//            actual k = k0 + 4q + s in slot q (any k permutation is legal as long as A and B agree);
//   TN form  C[m,n] = sum_k A[k,m] B[k,n]  (dW = X^T.dP, K = packed rows):  lane (li, q) loads the float4
//            A[k0 + 4q + s][m0 + 4 li .. +3] (16 lanes = one whole 256-byte tile row) and feeds FOUR 16-wide tiles (tile e
//            holds the rows 4c + e: a permutation of the output rows / columns that the epilogue undoes with float4 stores).
// Scheduling: a launch is a fixed grid; the iteration space of up to two products (tiles x k-steps, counted on the device
// from the actual extents) is cut into equal contiguous ranges, one per wave.  A tile cut between waves is finished by the
// wave that owns its first k-step: the others park their accumulators in a workspace slot with write-through (sc1)
// stores and raise a flag; the owner -- for which this tile is the LAST thing it does, while for the contributors it is
// the FIRST -- polls and sums them in a fixed order with sc1 loads (deterministic; no fences, no split-K slabs, no
// reduction launch).  The dW product writes the per-view weight gradients directly (column re-layout in its epilogue).
#include <stdlib.h>

#include <algorithm>

#include "common.h"
#include "kernels.h"

namespace eagcn {

typedef __attribute__((address_space(1))) unsigned gu32;
typedef __attribute__((address_space(1))) unsigned long long gu64;
#define EAGCN_RLX_AGENT __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT

__device__ int g_gemm3_timeouts = 0;       // hand-offs that gave up waiting (must stay 0; diagnostics: eagcn_gemm_sk_timeouts)
// The STICKY error word lives in host memory mapped into the device's address space: a kernel whose owner wave gave up
// waiting stores 1 there (system scope) and poisons its tile with NaN, and every later API call -- and the graph-replay
// host loop, which makes no API call per step -- reads the word without any synchronisation and fails loudly.
static int* g_g3_err_host = nullptr;       // host view
static int* g_g3_err_dev = nullptr;        // device view of the same word
static int* g3_err_word() {
    if (!g_g3_err_dev) {
        void* h = nullptr;
        if (hipHostMalloc(&h, 64, hipHostMallocMapped) != hipSuccess) return nullptr;
        memset(h, 0, 64);
        void* d = nullptr;
        if (hipHostGetDevicePointer(&d, h, 0) != hipSuccess) return nullptr;
        g_g3_err_host = (int*)h;
        g_g3_err_dev = (int*)d;
    }
    return g_g3_err_dev;
}

constexpr int G3_T = 64;                   // wave tile (both dimensions)
constexpr int G3_BK = 16;
constexpr int G3_SLOT = G3_T * G3_T;       // floats of one parked partial tile
constexpr int G3_MAX_RANGES = 8192;
#ifndef G3_FAST_LOOP
#define G3_FAST_LOOP 1
#endif
#ifndef G3_WAVES_PER_SIMD
#define G3_WAVES_PER_SIMD 1
#endif

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

struct G3Sched {
    int total, R, per, rem;                // R ranges; range l = [l*per + min(l,rem), ...)
    __device__ __forceinline__ int start(int l) const { return l * per + min(l, rem); }
};

__device__ __forceinline__ f32x4 g3_sel(bool ok, f32x4 v) { return ok ? v : (f32x4){0.f, 0.f, 0.f, 0.f}; }

// k-steps [kb, ke) of output tile `tile`.  mode 0: store, 1: park the partial + flag, 2: owner (add the contributors, store)
template <bool TN, bool SCATTER>
__device__ __forceinline__ void g3_segment(const G2Prob& p, const DwScatter& sc, const int Mx, const int Kx, const int tile,
                                           const int kb, const int ke, const int mode, const int l, const int tile_end_global,
                                           const G3Sched& sched, float* __restrict__ ws, unsigned* __restrict__ flags,
                                           int* __restrict__ err) {
    const int lane = threadIdx.x & 63;
    const int li = lane & 15, q = lane >> 4;
    const int gx = (p.N + G3_T - 1) / G3_T;
    const int ty = tile / gx, tx = tile - ty * gx;
    const int m0 = ty * G3_T, n0 = tx * G3_T;

    // operand addressing: NT: one row pointer per 16-row block; TN: one column offset, rows advance with k
    const float* ap[4];
    const float* bp[4];
    if constexpr (!TN) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            ap[i] = p.A + (size_t)max(min(m0 + 16 * i + li, Mx - 1), 0) * p.lda + 4 * q;
            // (tile j of the B side takes the rows 4 li + j, so that a lane's four results of a row are one float4)
            bp[i] = p.B + (size_t)max(min(n0 + 4 * li + i, p.N - 1), 0) * p.ldb + 4 * q;
        }
    } else {
        const float* a0 = p.A + max(min(m0 + 4 * li, p.M - 4), 0);
        const float* b0 = p.B + max(min(n0 + 4 * li, p.N - 4), 0);
#pragma unroll
        for (int s = 0; s < 4; ++s) { ap[s] = a0; bp[s] = b0; }
    }
    // two register sets: while the 64 MFMAs of one k-step run, the loads of the next step are in flight.  Only the LAST
    // k-step of a tile can reach beyond K (K is a multiple of 4, not of 16): the steady-state loop multiplies the loaded
    // registers as they are, the final step of the segment zeroes the out-of-range k by selects.
    auto load_step = [&](int it, f32x4 (&a)[4], f32x4 (&b)[4]) __attribute__((always_inline)) {
        const int k0 = it * G3_BK;
        if constexpr (!TN) {
            const int kc = max(min(k0 + 4 * q, Kx - 4), 0) - 4 * q;  // (the pointers already carry + 4q)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                a[i] = *reinterpret_cast<const f32x4*>(ap[i] + kc);
                b[i] = *reinterpret_cast<const f32x4*>(bp[i] + kc);
            }
        } else {
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const size_t kr = (size_t)max(min(k0 + 4 * q + s, Kx - 1), 0);
                a[s] = *reinterpret_cast<const f32x4*>(ap[s] + kr * p.lda);
                b[s] = *reinterpret_cast<const f32x4*>(bp[s] + kr * p.ldb);
            }
        }
    };

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // NT: a[i] = A[row block i][k0+4q .. +3], b[j] likewise: MFMA step s takes component s of both.
    // TN: a[s] = A[k0+4q+s][m0+4li .. +3]: component e is the operand of row-tile e; b[s] likewise for column-tile e.
    auto mma_step = [&](const f32x4 (&a)[4], const f32x4 (&b)[4]) __attribute__((always_inline)) {
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if constexpr (!TN) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i][s], b[j][s], acc[i][j], 0, 0, 0);
                    else acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s][i], b[s][j], acc[i][j], 0, 0, 0);
                }
    };
    auto mma_tail = [&](int it, f32x4 (&a)[4], f32x4 (&b)[4]) __attribute__((always_inline)) {
        const int k0 = it * G3_BK;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const bool v = TN ? (k0 + 4 * q + u < Kx) : (k0 + 4 * q < Kx);
            a[u] = g3_sel(v, a[u]);
            b[u] = g3_sel(v, b[u]);
        }
        mma_step(a, b);
    };

    // ring of four register sets: the loads of step t+3 are issued right before the MFMAs of step t, i.e. three steps
    // (3 x 64 MFMAs) of cover for the memory latency of a wave that has its SIMD to itself
    f32x4 sa[4][4], sb[4][4];
    int it = kb;
    load_step(min(it, ke - 1), sa[0], sb[0]);
    load_step(min(it + 1, ke - 1), sa[1], sb[1]);
    load_step(min(it + 2, ke - 1), sa[2], sb[2]);
#define EAGCN_G3_STEP(CUR, NXT)                                   \
    load_step(min(it + 3, ke - 1), sa[NXT], sb[NXT]);             \
    __builtin_amdgcn_sched_barrier(0);                            \
    mma_step(sa[CUR], sb[CUR]);                                   \
    __builtin_amdgcn_sched_barrier(0);                            \
    ++it;
#if G3_FAST_LOOP
    // Steady state proper: as long as the prefetched step (it + 3 .. it + 6 inside a group of four) is not the LAST step of the
    // segment nothing can reach beyond K or the segment, so the loads need no clamps at all -- per-lane pointers that are
    // bumped once per group (NT: the four steps of a group sit at constant byte offsets 0 / 64 / 128 / 192 behind them) or once
    // per step (TN: 16 rows further), and the eight loads of a step are spread over its 64 MFMAs by the scheduler directives
    // instead of forming a block of their own in front of them (the matrix pipe drained during every such block: the wave's
    // non-MFMA issue time was ~500 of ~2550 cycles per step).
    if (it + 7 < ke) {
        const float* fa[4];
        const float* fb[4];
        size_t stepA = 0, stepB = 0;
        if constexpr (!TN) {
#pragma unroll
            for (int i = 0; i < 4; ++i) { fa[i] = ap[i] + (size_t)(it + 3) * G3_BK; fb[i] = bp[i] + (size_t)(it + 3) * G3_BK; }
        } else {
            stepA = (size_t)G3_BK * p.lda;
            stepB = (size_t)G3_BK * p.ldb;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const size_t kr = (size_t)((it + 3) * G3_BK + 4 * q + u);
                fa[u] = ap[u] + kr * p.lda;
                fb[u] = bp[u] + kr * p.ldb;
            }
        }
#define EAGCN_G3_FSTEP(CUR, NXT, U)                                                                         \
    _Pragma("unroll") for (int e = 0; e < 4; ++e) {                                                         \
        if constexpr (!TN) {                                                                                \
            sa[NXT][e] = *reinterpret_cast<const f32x4*>(fa[e] + (U) * G3_BK);                              \
            sb[NXT][e] = *reinterpret_cast<const f32x4*>(fb[e] + (U) * G3_BK);                              \
        } else {                                                                                            \
            sa[NXT][e] = *reinterpret_cast<const f32x4*>(fa[e]);                                            \
            sb[NXT][e] = *reinterpret_cast<const f32x4*>(fb[e]);                                            \
            fa[e] += stepA;                                                                                 \
            fb[e] += stepB;                                                                                 \
        }                                                                                                   \
    }                                                                                                       \
    mma_step(sa[CUR], sb[CUR]);                                                                             \
    _Pragma("unroll") for (int g = 0; g < 8; ++g) {                                                         \
        __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);                                                  \
        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);                                                  \
        __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);                                                  \
        __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);                                                  \
    }                                                                                                       \
    __builtin_amdgcn_sched_barrier(0);                                                                      \
    ++it;
        while (it + 7 < ke) {
            EAGCN_G3_FSTEP(0, 3, 0)
            EAGCN_G3_FSTEP(1, 0, 1)
            EAGCN_G3_FSTEP(2, 1, 2)
            EAGCN_G3_FSTEP(3, 2, 3)
            if constexpr (!TN) {
#pragma unroll
                for (int i = 0; i < 4; ++i) { fa[i] += 4 * G3_BK; fb[i] += 4 * G3_BK; }
            }
        }
#undef EAGCN_G3_FSTEP
    }
#endif
    while (it + 4 < ke) {                            // the four steps it .. it+3 are not the last one of the segment
        EAGCN_G3_STEP(0, 3)
        EAGCN_G3_STEP(1, 0)
        EAGCN_G3_STEP(2, 1)
        EAGCN_G3_STEP(3, 2)
    }
    {
        const int r = ke - it;                       // 1 .. 4 steps left; step `it` sits in set 0
        if (r == 4) { EAGCN_G3_STEP(0, 3) } else if (r >= 2) { mma_step(sa[0], sb[0]); ++it; }
        if (r >= 3) { mma_step(sa[1], sb[1]); ++it; }
        if (r == 4) { mma_step(sa[2], sb[2]); ++it; }
        if (r == 1) mma_tail(it, sa[0], sb[0]);
        else if (r == 2) mma_tail(it, sa[1], sb[1]);
        else if (r == 3) mma_tail(it, sa[2], sb[2]);
        else mma_tail(it, sa[3], sb[3]);
    }
#undef EAGCN_G3_STEP

    // ---- hand-off of partial tiles (per wave; 16-byte write-through (sc1) stores and sc1 loads through a buffer
    //      descriptor -- no fences: sc1 stores leave the XCD's L2, sc1 loads bypass the CU's L1; 8-byte agent-scope
    //      atomics, the only other fence-free form, move the same bytes at about half the rate) ------------------------------
    if (mode == 1) {
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(ws + (size_t)l * G3_SLOT), 0, G3_SLOT * 4, 0x00020000);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, acc[i][j]), rs, ((i * 4 + j) * 64 + lane) * 16, 0, 16);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");              // this wave's stores have left
        if (lane == 0) __hip_atomic_store((gu32*)(flags + l), 1u, EAGCN_RLX_AGENT);
        return;
    }
    if (mode == 2) {
        // contributors: the ranges l+1, l+2, ... that start inside this tile (empty ranges export nothing); taken two at a
        // time so that the loads of two parked tiles are in flight together (the ring registers are free by now); the
        // additions stay in range order
        int last = l;
        while (last + 1 < sched.R && sched.start(last + 1) < tile_end_global) ++last;
        // A contributor that never shows up (the scheme needs every wave of the grid co-resident with its owner) must
        // not yield a silently wrong tile: the owner gives up after 2^22 polls, raises the sticky host-visible error word
        // and poisons its tile with NaN -- it never continues with a partial sum.
        bool timed_out = false;
        auto wait_flag = [&](int c) __attribute__((always_inline)) {
            unsigned spins = 0;
            while (__hip_atomic_load((gu32*)(flags + c), EAGCN_RLX_AGENT) == 0u) {
                __builtin_amdgcn_s_sleep(2);
                if (++spins > (1u << 22)) {
                    if (lane == 0) {
                        atomicAdd(&g_gemm3_timeouts, 1);
                        if (err) __hip_atomic_store(err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    }
                    timed_out = true;
                    break;
                }
            }
        };
        int c = l + 1;
        while (c <= last) {
            while (c <= last && sched.start(c + 1) <= sched.start(c)) ++c;
            if (c > last) break;
            const int c1 = c++;
            while (c <= last && sched.start(c + 1) <= sched.start(c)) ++c;
            const int c2 = c <= last ? c++ : -1;
            wait_flag(c1);
            if (c2 >= 0) wait_flag(c2);
            const __amdgpu_buffer_rsrc_t r1 = __builtin_amdgcn_make_buffer_rsrc((void*)(ws + (size_t)c1 * G3_SLOT), 0, G3_SLOT * 4, 0x00020000);
            const __amdgpu_buffer_rsrc_t r2 = __builtin_amdgcn_make_buffer_rsrc((void*)(ws + (size_t)(c2 >= 0 ? c2 : c1) * G3_SLOT), 0, G3_SLOT * 4, 0x00020000);
            u32x4 t1[16], t2[16];
#pragma unroll
            for (int e = 0; e < 16; ++e) t1[e] = __builtin_amdgcn_raw_buffer_load_b128(r1, (e * 64 + lane) * 16, 0, 16);
            if (c2 >= 0) {
#pragma unroll
                for (int e = 0; e < 16; ++e) t2[e] = __builtin_amdgcn_raw_buffer_load_b128(r2, (e * 64 + lane) * 16, 0, 16);
            }
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[e >> 2][e & 3] += __builtin_bit_cast(f32x4, t1[e]);
            if (c2 >= 0) {
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[e >> 2][e & 3] += __builtin_bit_cast(f32x4, t2[e]);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (lane == 0) {
                __hip_atomic_store((gu32*)(flags + c1), 0u, EAGCN_RLX_AGENT);
                if (c2 >= 0) __hip_atomic_store((gu32*)(flags + c2), 0u, EAGCN_RLX_AGENT);
            }
        }
        if (timed_out) {
            const float qnan = __builtin_nanf("");
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[e >> 2][e & 3] = (f32x4){qnan, qnan, qnan, qnan};
        }
    }

    // ---- epilogue: D layout col c = lane & 15, row rho = 4 (lane >> 4) + reg -----------------------------------------------
    if constexpr (!TN) {
        const bool vec_c = (p.ldc & 3) == 0 && (reinterpret_cast<uintptr_t>(p.C) & 15) == 0;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = m0 + 16 * i + 4 * q + r;
                const int col = n0 + 4 * li;                           // tile j holds the columns 4 c + j
                if (row >= Mx || col >= p.N) continue;
                float* d = p.C + (size_t)row * p.ldc + col;
                if (vec_c && col + 3 < p.N) {
                    *reinterpret_cast<f32x4*>(d) = (f32x4){acc[i][0][r], acc[i][1][r], acc[i][2][r], acc[i][3][r]};
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (col + j < p.N) d[j] = acc[i][j][r];
                }
            }
    } else {
        const int col = n0 + 4 * li;                                   // tile j holds the columns 4 c + j: one float4
        // dW of a layer (SCATTER): row = packed input column, col = packed output column -> blockK.graph_conv.weight.grad.
        // This runs at the very end of a tile's serial import chain, so it is kept short: the column side (view, width,
        // target) is resolved once per lane, the row side is the identity whenever the input layout has no padded
        // columns (DwScatter::in_identity: widths that are multiples of 16, every hidden layer of the stated configs),
        // and a lane's four results leave as ONE 16-byte store when the view's width allows it.
        int off_k = 0, wk = 0, f = 0;
        float* dwk = nullptr;
        bool vec = false;
        if constexpr (SCATTER) {
            int k = 0;
#pragma unroll
            for (int vv = 1; vv < EAGCN_MAX_VIEWS; ++vv) k += (vv < sc.vc.K && col >= sc.vc.off[vv]) ? 1 : 0;
#pragma unroll
            for (int vv = 0; vv < EAGCN_MAX_VIEWS; ++vv)
                if (vv == k) { off_k = sc.vc.off[vv]; wk = sc.vc.width[vv]; dwk = sc.dW[vv]; }
            f = col - off_k;
            vec = (wk & 3) == 0 && f + 3 < wk && (reinterpret_cast<uintptr_t>(dwk) & 15) == 0;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = m0 + 4 * (4 * q + r) + i;              // tile i holds the rows 4 rho + i
                if (row >= Mx || col >= p.N) continue;                 // (N is a multiple of 4)
                const f32x4 v = (f32x4){acc[i][0][r], acc[i][1][r], acc[i][2][r], acc[i][3][r]};
                if constexpr (SCATTER) {
                    int fi = row;
                    if (!sc.in_identity) {
                        int eo = 0, po = 0;
                        bool done = false;
                        fi = -1;
#pragma unroll
                        for (int sg = 0; sg < EAGCN_MAX_SEGS; ++sg) {
                            if (sg < sc.in.nseg && !done) {
                                if (row < po + sc.in.p[sg]) { fi = (row - po < sc.in.w[sg]) ? eo + (row - po) : -1; done = true; }
                                eo += sc.in.w[sg];
                                po += sc.in.p[sg];
                            }
                        }
                    }
                    if (fi >= 0 && f < wk) {
                        float* d = dwk + (size_t)fi * wk + f;
                        if (vec) {
                            *reinterpret_cast<f32x4*>(d) = v;
                        } else {
#pragma unroll
                            for (int e = 0; e < 4; ++e)
                                if (f + e < wk) d[e] = v[e];
                        }
                    }
                } else {
                    *reinterpret_cast<f32x4*>(p.C + (size_t)row * p.ldc + col) = v;
                }
            }
    }
}

// XCD-aware logical index: dispatch slot b runs on XCD b % 8; give every XCD a contiguous run of ranges
__device__ __forceinline__ int g3_logical(int b, int G) {
    const int xcd = b & 7, qn = G >> 3, rn = G & 7;
    return (xcd < rn ? xcd * (qn + 1) : rn * (qn + 1) + (xcd - rn) * qn) + (b >> 3);
}

// problem 0: NT or TN (TN0); optional problem 1 is always TN (the dW that rides with a dX), optionally scattered.
// XK (pairs only): XCD-local schedule (kernels.h g3_plan): dW leaves as one partial slab per segment at p1.C + x * xk_slab.
template <bool TN0, bool HAS1, bool SCAT, bool XK>
__global__ __launch_bounds__(256, G3_WAVES_PER_SIMD) void gemm3_kernel(G2Prob p0, G2Prob p1, DwScatter sc, float* __restrict__ ws,
                                                    unsigned* __restrict__ flags, int* __restrict__ err, size_t xk_slab) {
    const int G = gridDim.x;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int l = g3_logical(blockIdx.x, G) * 4 + wave;
    const int M0 = p0.M_dev ? min(*p0.M_dev, p0.M) : p0.M;
    const int K0 = p0.K_dev ? min(*p0.K_dev, p0.K) : p0.K;
    const int ipt0 = max(1, (K0 + G3_BK - 1) / G3_BK);
    const int it0 = ((M0 + G3_T - 1) / G3_T) * ((p0.N + G3_T - 1) / G3_T) * ipt0;
    int M1 = 0, K1 = 0, ipt1 = 1, it1 = 0;
    if constexpr (HAS1) {
        M1 = p1.M_dev ? min(*p1.M_dev, p1.M) : p1.M;
        K1 = p1.K_dev ? min(*p1.K_dev, p1.K) : p1.K;
        ipt1 = max(1, (K1 + G3_BK - 1) / G3_BK);
        it1 = ((M1 + G3_T - 1) / G3_T) * ((p1.N + G3_T - 1) / G3_T) * ipt1;
    }
    G3Sched sched;
    sched.total = it0 + it1;
    sched.R = 4 * G;
    sched.per = sched.total / sched.R;
    sched.rem = sched.total - sched.per * sched.R;
    int it = sched.start(l);
    const int end = sched.start(l + 1);
    if constexpr (XK && HAS1 && !TN0) {
        const int RB = (M0 + G3_T - 1) / G3_T, CT0 = (p0.N + G3_T - 1) / G3_T;
        const int T1 = ((M1 + G3_T - 1) / G3_T) * ((p1.N + G3_T - 1) / G3_T);
        G3Plan pl;
        g3_plan(RB, CT0, ipt0, T1, ipt1, G, pl);
        const int rowit = CT0 * ipt0;                                   // iterations of one dX row block
        while (it < end) {
            // segment of `it`: the last x with base_x <= it (empty segments share their base with the next one)
            int ax = 0, ax1 = pl.a[1], cx = 0, cx1 = pl.c[1], x = 0;
#pragma unroll
            for (int y = 1; y < G3_XSEG; ++y) {
                const bool in = it >= pl.a[y] * rowit + T1 * pl.c[y];
                ax = in ? pl.a[y] : ax; ax1 = in ? pl.a[y + 1] : ax1;
                cx = in ? pl.c[y] : cx; cx1 = in ? pl.c[y + 1] : cx1;
                x = in ? y : x;
            }
            const int base = ax * rowit + T1 * cx;
            const int nA = (ax1 - ax) * rowit;
            const int local = it - base;
            if (local < nA) {                                           // dX tile of this segment's row blocks
                const int tl = local / ipt0, kb = local - tl * ipt0;
                const int run_end = base + (tl + 1) * ipt0;
                const int seg_end = min(end, run_end);
                const int mode = kb > 0 ? 1 : (seg_end < run_end ? 2 : 0);
                g3_segment<false, false>(p0, sc, M0, K0, ax * CT0 + tl, kb, kb + (seg_end - it), mode, l, run_end, sched, ws, flags, err);
                it = seg_end;
            } else {                                                    // k-steps [cx, cx1) of a dW tile -> partial slab x
                const int kw = cx1 - cx, l2 = local - nA;
                const int tile = l2 / kw, kk = l2 - tile * kw;
                const int run_end = base + nA + (tile + 1) * kw;
                const int seg_end = min(end, run_end);
                const int mode = kk > 0 ? 1 : (seg_end < run_end ? 2 : 0);
                G2Prob px = p1;
                px.C = p1.C + (size_t)x * xk_slab;
                g3_segment<true, false>(px, sc, M1, K1, tile, cx + kk, cx + kk + (seg_end - it), mode, l, run_end, sched, ws, flags, err);
                it = seg_end;
            }
        }
        return;
    }
    while (it < end) {
        const bool second = HAS1 && it >= it0;
        const int base = second ? it0 : 0;
        const int ipt = second ? ipt1 : ipt0;
        const int local = it - base;
        const int tile = local / ipt;
        const int kb = local - tile * ipt;
        const int tile_end = base + (tile + 1) * ipt;
        const int seg_end = min(end, tile_end);
        const int ke = kb + (seg_end - it);
        const int mode = kb > 0 ? 1 : (seg_end < tile_end ? 2 : 0);
        if (!second) g3_segment<TN0, SCAT && !HAS1>(p0, sc, M0, K0, tile, kb, ke, mode, l, tile_end, sched, ws, flags, err);
        else if constexpr (HAS1) g3_segment<true, SCAT>(p1, sc, M1, K1, tile, kb, ke, mode, l, tile_end, sched, ws, flags, err);
        it = seg_end;
    }
}

// ---- host side ---------------------------------------------------------------------------------------------------------
static int g3_env(const char* name, int dflt) {
    const char* e = getenv(name);
    return e ? atoi(e) : dflt;
}
int gemm3_grid() {           // workgroups of four autonomous waves: one wave per SIMD (two measured 40 % slower: the fragment
                             // streams of eight waves no longer fit the CU's 32 KB vector cache)
    // default: one workgroup per CU of THIS device (the hand-off needs the whole grid co-resident; 256 on an MI355X)
    static const int g = [] {
        int cus = 256, dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
            cus = prop.multiProcessorCount;
        (void)hipGetLastError();               // (no device in the build container: keep the nominal 256)
        int v = g3_env("EAGCN_GEMM3_WGS", cus);
        return std::max(8, std::min(v, G3_MAX_RANGES / 4));
    }();
    return g;
}
size_t gemm3_workspace_bytes() {       // one parked tile + one flag per wave of the grid
    const size_t ranges = (size_t)gemm3_grid() * 4;
    return align256(ranges * G3_SLOT * sizeof(float)) + align256(ranges * sizeof(unsigned));
}
static bool g3_aligned(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// NT: ta = 0, tb = 1; TN: ta = 1, tb = 0.  float4 loads without predication: aligned operands, whole runs
bool gemm3_ok(const GemmDesc& g) {
    if (g.M <= 0 || g.N <= 0 || g.K <= 0) return false;
    if (!g3_aligned(g.A) || !g3_aligned(g.B) || (g.lda & 3) || (g.ldb & 3)) return false;
    if ((double)cdiv(g.M, G3_T) * cdiv(g.N, G3_T) * cdiv(g.K, G3_BK) >= 1.0e9) return false;
    if (g.ta == 0 && g.tb == 1) return (g.K & 3) == 0 && g.K >= 4 && !g.K_dev;
    if (g.ta == 1 && g.tb == 0) return (g.M & 3) == 0 && (g.N & 3) == 0 && !g.M_dev && g.M >= 4 && g.N >= 4 && (g.ldc & 3) == 0 &&
                                       g3_aligned(g.C);
    return false;
}

static G2Prob g3_prob(const GemmDesc& g) {
    G2Prob p;
    p.A = g.A; p.B = g.B; p.C = g.C; p.lda = g.lda; p.ldb = g.ldb; p.ldc = g.ldc;
    p.M = g.M; p.N = g.N; p.K = g.K; p.M_dev = g.M_dev; p.K_dev = g.K_dev;
    return p;
}

static int g3_prepare(void* workspace, size_t bytes, float** ws, unsigned** flags) {
    EAGCN_CHECK_ARG(workspace && bytes >= gemm3_workspace_bytes(), "gemm: workspace too small (%zu < %zu)", bytes, gemm3_workspace_bytes());
    EAGCN_CHECK_ARG(g3_aligned(workspace), "gemm: workspace must be 16-byte aligned");
    *ws = (float*)workspace;
    const size_t ranges = (size_t)gemm3_grid() * 4;
    *flags = (unsigned*)((char*)workspace + align256(ranges * G3_SLOT * sizeof(float)));
    return EAGCN_OK;
}
__global__ void g3_zero_kernel(unsigned* __restrict__ p, int n, double* __restrict__ z, int nz) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = 0u;
    if (i < nz) z[i] = 0.0;
}
// words to clear from the first hand-off flag on: the flags, and -- when the caller's region continues with the edge-gradient
// accumulators of kernels.h (every layer scratch carving does) -- those as well
static int g3_clear_words(size_t bytes) {
    const size_t ranges = (size_t)gemm3_grid() * 4;
    size_t n = align256(ranges * sizeof(unsigned));
    if (bytes >= gemm3_workspace_bytes() + edge_acc_bytes()) n += edge_acc_bytes();
    return (int)(n / sizeof(unsigned));
}
int gemm3_clear_flags(void* workspace, size_t bytes, hipStream_t s, double* zero, int nzero) {
    float* ws; unsigned* flags;
    int rc = g3_prepare(workspace, bytes, &ws, &flags);
    if (rc) return rc;
    // a KERNEL, not hipMemsetAsync: as a memset node of a captured graph the clear was not ordered with the kernel nodes
    // around it on replay (ROCm 7.2: owners then read parked tiles of an earlier launch; tests/probe_graph_replay.py)
    const int n = g3_clear_words(bytes);
    g3_zero_kernel<<<cdiv(std::max(n, nzero), 256), 256, 0, s>>>(flags, n, zero, zero ? nzero : 0);
    EAGCN_LAUNCH_CHECK();
    return EAGCN_OK;
}

// Zero-fill by a KERNEL (see gemm3_clear_flags: a memset node of a captured graph is not ordered with its neighbours on replay):
// every clear that can end up inside a captured sequence goes through here.  p 4-byte aligned, bytes a multiple of 4.
__global__ void zero_words_kernel(unsigned* __restrict__ p, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = 0u;
}
int zero_fill(void* p, size_t bytes, hipStream_t s) {
    if (bytes == 0) return EAGCN_OK;
    EAGCN_CHECK_ARG(p && (bytes & 3) == 0 && (reinterpret_cast<uintptr_t>(p) & 3) == 0, "zero_fill: unaligned region");
    const size_t n = bytes / 4;
    zero_words_kernel<<<(unsigned)std::max<size_t>(1, std::min<size_t>((n + 255) / 256, 4096)), 256, 0, s>>>((unsigned*)p, n);
    EAGCN_LAUNCH_CHECK();
    return EAGCN_OK;
}

int gemm3_zero_job(void* workspace, size_t bytes, double* zero, int nzero, ZeroJob* out) {
    float* ws; unsigned* flags;
    int rc = g3_prepare(workspace, bytes, &ws, &flags);
    if (rc) return rc;
    out->u = flags; out->nu = g3_clear_words(bytes); out->d = zero; out->nd = zero ? nzero : 0;
    return EAGCN_OK;
}

// one product; with `sc0` (TN only) the result goes to the per-view weight gradients instead of g.C
int launch_gemm3(const GemmDesc& g, const DwScatter* sc0, void* workspace, size_t bytes, hipStream_t s) {
    EAGCN_CHECK_ARG(gemm3_ok(g), "gemm: operands not 16-byte aligned / extents not multiples of 4 / unsupported form");
    EAGCN_CHECK_ARG(!sc0 || g.ta == 1, "gemm: the weight-gradient epilogue belongs to the TN form");
    float* ws; unsigned* flags;
    int rc = g3_prepare(workspace, bytes, &ws, &flags);
    if (rc) return rc;
    DwScatter sc;
    if (sc0) sc = *sc0; else memset(&sc, 0, sizeof(sc));
    sc.in_identity = 1;
    for (int i = 0; i < sc.in.nseg; ++i) sc.in_identity &= sc.in.w[i] == sc.in.p[i] ? 1 : 0;
    const G2Prob p = g3_prob(g);
    const int G = gemm3_grid();
    ProfScope ps(g.prof_tag, s, g.work > 0.0 ? g.work : 2.0 * g.M * g.N * g.K);
    if (g.ta == 0) gemm3_kernel<false, false, false, false><<<G, 256, 0, s>>>(p, p, sc, ws, flags, g3_err_word(), 0);
    else if (sc0) gemm3_kernel<true, false, true, false><<<G, 256, 0, s>>>(p, p, sc, ws, flags, g3_err_word(), 0);
    else gemm3_kernel<true, false, false, false><<<G, 256, 0, s>>>(p, p, sc, ws, flags, g3_err_word(), 0);
    EAGCN_LAUNCH_CHECK();
    return EAGCN_OK;
}

// dX = dP.W^T (NT) and dW = X^T.dP (TN) of a layer in one balanced launch
bool gemm3_xk_enabled() {
    // measured on MI355X (gpurun_out/r3a): 8x less fabric traffic and NOT faster (pair 77 vs 71 us stand-alone at 4809
    // rows, 259 vs 233 us at 19200) -- the kernel is not bound by operand fetch; off unless EAGCN_GEMM3_XK=1
    static const bool on = g3_env("EAGCN_GEMM3_XK", 0) != 0;
    return on;
}

int launch_gemm3_pair(const GemmDesc& dx, const GemmDesc& dw, const DwScatter* sc0, void* workspace, size_t bytes, hipStream_t s,
                      size_t xk_slab) {
    EAGCN_CHECK_ARG(gemm3_ok(dx) && gemm3_ok(dw) && dx.ta == 0 && dw.ta == 1, "gemm pair: unsupported operands");
    EAGCN_CHECK_ARG(xk_slab == 0 || !sc0, "gemm pair: the XCD-local schedule writes partial slabs, not scattered gradients");
    float* ws; unsigned* flags;
    int rc = g3_prepare(workspace, bytes, &ws, &flags);
    if (rc) return rc;
    DwScatter sc;
    if (sc0) sc = *sc0; else memset(&sc, 0, sizeof(sc));
    sc.in_identity = 1;
    for (int i = 0; i < sc.in.nseg; ++i) sc.in_identity &= sc.in.w[i] == sc.in.p[i] ? 1 : 0;
    const int G = gemm3_grid();
    const double w0 = dx.work > 0.0 ? dx.work : 2.0 * dx.M * dx.N * dx.K;
    const double w1 = dw.work > 0.0 ? dw.work : 2.0 * dw.M * dw.N * dw.K;
    ProfScope ps(PROF_GEMM_PAIR, s, w0 + w1);
    if (xk_slab) gemm3_kernel<false, true, false, true><<<G, 256, 0, s>>>(g3_prob(dx), g3_prob(dw), sc, ws, flags, g3_err_word(), xk_slab);
    else if (sc0) gemm3_kernel<false, true, true, false><<<G, 256, 0, s>>>(g3_prob(dx), g3_prob(dw), sc, ws, flags, g3_err_word(), 0);
    else gemm3_kernel<false, true, false, false><<<G, 256, 0, s>>>(g3_prob(dx), g3_prob(dw), sc, ws, flags, g3_err_word(), 0);
    EAGCN_LAUNCH_CHECK();
    return EAGCN_OK;
}

}  // namespace eagcn

using namespace eagcn;

extern "C" size_t eagcn_gemm_sk_workspace_bytes(void) { return gemm3_workspace_bytes(); }

/* segment boundaries of the XCD-local paired schedule for (M0,N0,K0) x (M1,N1,K1) on a grid of `wgs` workgroups (0: the
 * library's grid): a[9] row-block boundaries of the first product, c[9] k-step boundaries of the second (host-side mirror of
 * what the kernel computes from the device-side extents; lets the CPU tests check the partition) */
extern "C" int eagcn_gemm_sk_plan(int M0, int N0, int K0, int M1, int N1, int K1, int wgs, int* a, int* c) {
    EAGCN_CHECK_ARG(a && c, "eagcn_gemm_sk_plan: null output");
    G3Plan pl;
    g3_plan(cdiv(M0, G3_T), cdiv(N0, G3_T), std::max(1, cdiv(K0, G3_BK)), cdiv(M1, G3_T) * cdiv(N1, G3_T), std::max(1, cdiv(K1, G3_BK)),
            wgs > 0 ? wgs : gemm3_grid(), pl);
    for (int x = 0; x <= G3_XSEG; ++x) { a[x] = pl.a[x]; c[x] = pl.c[x]; }
    return EAGCN_OK;
}

int eagcn::gemm3_failed() { return g_g3_err_host ? __atomic_load_n(g_g3_err_host, __ATOMIC_RELAXED) : 0; }

/* number of hand-offs that gave up waiting since load (synchronising read of the device counter; must stay 0) */
extern "C" int eagcn_gemm_sk_timeouts(void) {
    int v = -1;
    if (hipMemcpyFromSymbol(&v, HIP_SYMBOL(g_gemm3_timeouts), sizeof(int)) != hipSuccess) return -1;
    return v;
}
/* sticky error word (host-mapped, no synchronisation): non-zero once any hand-off of any launch timed out -- the tile it
 * belonged to was poisoned with NaN.  Every model / layer entry point refuses to run while it is set. */
extern "C" int eagcn_gemm_sk_failed(void) { return gemm3_failed(); }
extern "C" void eagcn_gemm_sk_inject_failure(void) {     /* test hook: what a timed-out owner wave does to the sticky word */
    if (g3_err_word()) __atomic_store_n(g_g3_err_host, 1, __ATOMIC_RELAXED);
}
extern "C" void eagcn_gemm_sk_reset_failed(void) {
    if (g_g3_err_host) __atomic_store_n(g_g3_err_host, 0, __ATOMIC_RELAXED);
    int z = 0;
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_gemm3_timeouts), &z, sizeof(int));
}

extern "C" int eagcn_gemm_f32_sk(int ta, int tb, int M, int N, int K, const float* A, int lda, const float* B, int ldb, float* C,
                                 int ldc, void* workspace, size_t workspace_bytes, void* stream) {
    EAGCN_CHECK_ARG(A && B && C, "eagcn_gemm_f32_sk: null operand");
    GemmDesc g{ta, tb, M, N, K, A, lda, B, ldb, C, ldc, 1, 0};
    int rc = gemm3_clear_flags(workspace, workspace_bytes, (hipStream_t)stream);
    return rc ? rc : launch_gemm3(g, nullptr, workspace, workspace_bytes, (hipStream_t)stream);
}

/* The same pair on the XCD-local schedule (kernels.h g3_plan): C1 receives G3_XSEG = 8 partial slabs of M1 x ldc1 floats,
 * `slab` floats apart (cleared first: a segment without k-steps of the second product writes nothing); their sum in slab
 * order is the product (test / benchmark entry). */
extern "C" int eagcn_gemm_pair_sk_slabs(int M0, int N0, int K0, const float* A0, int lda0, const float* B0, int ldb0, float* C0, int ldc0,
                                        int M1, int N1, int K1, const float* A1, int lda1, const float* B1, int ldb1, float* C1, int ldc1,
                                        size_t slab, void* workspace, size_t workspace_bytes, void* stream) {
    EAGCN_CHECK_ARG(A0 && B0 && C0 && A1 && B1 && C1, "eagcn_gemm_pair_sk_slabs: null operand");
    EAGCN_CHECK_ARG(slab >= (size_t)M1 * ldc1, "eagcn_gemm_pair_sk_slabs: slab stride smaller than one matrix");
    GemmDesc g0{0, 1, M0, N0, K0, A0, lda0, B0, ldb0, C0, ldc0, 1, 0};
    GemmDesc g1{1, 0, M1, N1, K1, A1, lda1, B1, ldb1, C1, ldc1, 1, 0};
    { int rcz = zero_fill(C1, slab * G3_XSEG * sizeof(float), (hipStream_t)stream); if (rcz) return rcz; }
    int rc = gemm3_clear_flags(workspace, workspace_bytes, (hipStream_t)stream);
    return rc ? rc : launch_gemm3_pair(g0, g1, nullptr, workspace, workspace_bytes, (hipStream_t)stream, slab);
}

/* dX[M0,N0] = A0[M0,K0] . B0[N0,K0]^T  and  dW[M1,N1] = A1[K1,M1]^T . B1[K1,N1] in ONE launch (test / benchmark entry of the
 * paired backward products of a layer) */
extern "C" int eagcn_gemm_pair_sk(int M0, int N0, int K0, const float* A0, int lda0, const float* B0, int ldb0, float* C0, int ldc0,
                                  int M1, int N1, int K1, const float* A1, int lda1, const float* B1, int ldb1, float* C1, int ldc1,
                                  void* workspace, size_t workspace_bytes, void* stream) {
    EAGCN_CHECK_ARG(A0 && B0 && C0 && A1 && B1 && C1, "eagcn_gemm_pair_sk: null operand");
    GemmDesc g0{0, 1, M0, N0, K0, A0, lda0, B0, ldb0, C0, ldc0, 1, 0};
    GemmDesc g1{1, 0, M1, N1, K1, A1, lda1, B1, ldb1, C1, ldc1, 1, 0};
    int rc = gemm3_clear_flags(workspace, workspace_bytes, (hipStream_t)stream);
    return rc ? rc : launch_gemm3_pair(g0, g1, nullptr, workspace, workspace_bytes, (hipStream_t)stream);
}
