// Internal helpers shared by the HIP translation units (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/eagcn_hip.h"

namespace eagcn {

constexpr float TINY = 1e-9f;   // reference layers.py:294
constexpr int WAVE = 64;

void set_error(const char* fmt, ...);

#define EAGCN_CHECK_ARG(cond, ...)                 \
    do {                                           \
        if (!(cond)) {                             \
            ::eagcn::set_error(__VA_ARGS__);       \
            return EAGCN_ERR_ARG;                  \
        }                                          \
    } while (0)

#define EAGCN_HIP(call)                                                                   \
    do {                                                                                  \
        hipError_t e_ = (call);                                                           \
        if (e_ != hipSuccess) {                                                           \
            ::eagcn::set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #call,              \
                               hipGetErrorString(e_));                                    \
            return EAGCN_ERR_HIP;                                                         \
        }                                                                                 \
    } while (0)

#define EAGCN_LAUNCH_CHECK() EAGCN_HIP(hipGetLastError())

inline int pad16(int w) { return (w + 15) & ~15; }
inline int pad4(int w) { return (w + 3) & ~3; }
inline int cdiv(int a, int b) { return (a + b - 1) / b; }
inline size_t align256(size_t x) { return (x + 255) & ~size_t(255); }

typedef float f32x4 __attribute__((ext_vector_type(4)));

// ---- device helpers ----------------------------------------------------------------------------
__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + __expf(-x)); }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ int wave_sum(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ int wave_max(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = max(v, __shfl_xor(v, o));
    return v;
}

// partial dot product of two rows over columns sl, sl+16, ... < n (16-lane group, lane sl): the loads of
// DOT_U column steps are issued together before any FMA, so a group has 2*DOT_U loads in flight instead
// of paying one memory round trip per step (measured: a rolled loop left edge_grad 84 % in s_waitcnt)
constexpr int DOT_U = 10;
__device__ __forceinline__ float dot16(const float* __restrict__ a, const float* __restrict__ b, int sl, int n) {
    float acc = 0.0f;
    for (int c0 = sl; c0 < n; c0 += 16 * DOT_U) {
        float x[DOT_U], y[DOT_U];
#pragma unroll
        for (int u = 0; u < DOT_U; ++u) {
            const int c = c0 + 16 * u;
            const bool ok = c < n;
            x[u] = ok ? a[c] : 0.0f;
            y[u] = ok ? b[c] : 0.0f;
        }
#pragma unroll
        for (int u = 0; u < DOT_U; ++u) acc += x[u] * y[u];
    }
    return acc;
}

// four partial dot products of row `a` with rows p[0..3] at once (same lane layout as dot16): the loads of
// `a` are shared and the four P rows are requested together, so several bonds of an atom cost ONE memory
// round trip instead of one each
__device__ __forceinline__ void dot16x4(const float* __restrict__ a, const float* const (&p)[4], int sl, int n,
                                        float (&out)[4]) {
    out[0] = out[1] = out[2] = out[3] = 0.0f;
    constexpr int U = 5;
    for (int c0 = sl; c0 < n; c0 += 16 * U) {
        float x[U], y[4][U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int c = c0 + 16 * u;
            const bool ok = c < n;
            x[u] = ok ? a[c] : 0.0f;
#pragma unroll
            for (int h = 0; h < 4; ++h) y[h][u] = ok ? p[h][c] : 0.0f;
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int h = 0; h < 4; ++h) out[h] += x[u] * y[h][u];
    }
}

// The same two dot products on 16-byte loads (rows that are 16-byte aligned and a multiple of 4 floats long: the padded view
// segments of the layer matrices): lane sl takes the float4s sl, sl+16, ... -- 16 lanes cover 256 contiguous bytes per load, a
// quarter of the memory instructions of the scalar forms above for the same bytes.
constexpr int DOTV_U = 3;
__device__ __forceinline__ float dot16v(const float* __restrict__ a, const float* __restrict__ b, int sl, int n) {
    const int nf4 = n >> 2;
    float acc = 0.0f;
    for (int c0 = sl; c0 < nf4; c0 += 16 * DOTV_U) {
        float4 x[DOTV_U], y[DOTV_U];
#pragma unroll
        for (int u = 0; u < DOTV_U; ++u) {
            const int c = c0 + 16 * u;
            const bool ok = c < nf4;
            x[u] = ok ? *reinterpret_cast<const float4*>(a + 4 * c) : make_float4(0.f, 0.f, 0.f, 0.f);
            y[u] = ok ? *reinterpret_cast<const float4*>(b + 4 * c) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int u = 0; u < DOTV_U; ++u) acc += (x[u].x * y[u].x + x[u].y * y[u].y) + (x[u].z * y[u].z + x[u].w * y[u].w);
    }
    return acc;
}
__device__ __forceinline__ void dot16x4v(const float* __restrict__ a, const float* const (&p)[4], int sl, int n, float (&out)[4]) {
    const int nf4 = n >> 2;
    out[0] = out[1] = out[2] = out[3] = 0.0f;
    constexpr int U = 2;
    for (int c0 = sl; c0 < nf4; c0 += 16 * U) {
        float4 x[U], y[4][U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int c = c0 + 16 * u;
            const bool ok = c < nf4;
            x[u] = ok ? *reinterpret_cast<const float4*>(a + 4 * c) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int h = 0; h < 4; ++h) y[h][u] = ok ? *reinterpret_cast<const float4*>(p[h] + 4 * c) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int h = 0; h < 4; ++h)
                out[h] += (x[u].x * y[h][u].x + x[u].y * y[h][u].y) + (x[u].z * y[h][u].z + x[u].w * y[h][u].w);
    }
}

// counter-based dropout stream: one 32-bit draw per (seed, element index)
__device__ __forceinline__ uint32_t rng_u32(uint64_t seed, uint64_t idx) {
    uint64_t z = idx * 0x9E3779B97F4A7C15ull + seed;
    z ^= z >> 32;
    z *= 0xD6E8FEB86659FD93ull;
    z ^= z >> 32;
    z *= 0xD6E8FEB86659FD93ull;
    z ^= z >> 32;
    return (uint32_t)z;
}
// 64 mixed bits per (seed, index): four 16-bit draws at once (dropout of the NON-STORED rows of a Weighted_sum layer, where
// only the NUMBER of kept rows per (molecule, view, column) matters: readout.hip)
__device__ __forceinline__ uint64_t rng_u64(uint64_t seed, uint64_t idx) {
    uint64_t z = idx * 0x9E3779B97F4A7C15ull + seed;
    z ^= z >> 32;
    z *= 0xD6E8FEB86659FD93ull;
    z ^= z >> 32;
    z *= 0xD6E8FEB86659FD93ull;
    z ^= z >> 32;
    return z;
}
constexpr uint64_t PAD_STREAM_BASE = 1ull << 40;     // index space of the non-stored rows' draws (rows beyond any packed row)
// keep-scale of an element: 0 (dropped) or 1/(1-p); thr = p * 2^32
__device__ __forceinline__ float drop_scale(uint64_t seed, uint64_t idx, uint32_t thr, float inv_keep) {
    return rng_u32(seed, idx) >= thr ? inv_keep : 0.0f;
}

// Dropout of the STORED rows of a graph-conv layer (element index r * Fp + column): one 64-bit hash serves FOUR adjacent
// elements -- element idx takes the 16-bit field idx & 3 of the draw of idx >> 2 and is kept when that field is >= p * 2^16 (the
// granularity of the non-stored rows' draws, readout.hip; p = 0.3 is held to 3e-6) -- because the passes that apply or
// regenerate the mask (bn_apply, the fused read-out, bn_bwd_reduce) were bound by the hash's 64-bit multiplies: a 64-bit
// multiply is four quarter-rate 32-bit ones, ~200 cycles per hash and wave.  thr = p * 2^32 as everywhere else.
__device__ __forceinline__ float drop_scale_el(uint64_t seed, uint64_t idx, uint32_t thr, float inv_keep) {
    const uint64_t z = rng_u64(seed, idx >> 2);
    return ((uint32_t)(z >> (16 * (int)(idx & 3))) & 0xFFFFu) >= (thr >> 16) ? inv_keep : 0.0f;
}
// four adjacent elements starting at an index that is a multiple of 4
__device__ __forceinline__ void drop_scale4(uint64_t seed, uint64_t idx, uint32_t thr, float inv_keep, float (&out)[4]) {
    const uint64_t z = rng_u64(seed, idx >> 2);
    const uint32_t lo = (uint32_t)z, hi = (uint32_t)(z >> 32), t16 = thr >> 16;
    out[0] = (lo & 0xFFFFu) >= t16 ? inv_keep : 0.0f;
    out[1] = (lo >> 16) >= t16 ? inv_keep : 0.0f;
    out[2] = (hi & 0xFFFFu) >= t16 ? inv_keep : 0.0f;
    out[3] = (hi >> 16) >= t16 ? inv_keep : 0.0f;
}

// ---- actual extents live in device memory ----------------------------------------------------------
// eagcn_batch.T / .n_tiles are CAPACITIES (buffer strides, grid sizing); the actual packed row count
// and tile count are read from meta[] on the device, so no launch depends on a host read-back.
__device__ __forceinline__ int dev_rows(const eagcn_batch& bt) { return min(bt.meta[EAGCN_META_T], bt.T); }
// padded molecule size of the batch as the reference sees it (<= the capacity bt.N): BatchNorm row counts, filler weights
__device__ __forceinline__ int dev_n(const eagcn_batch& bt) { const int v = bt.meta[EAGCN_META_NLOG]; return v > 0 ? min(v, bt.N) : bt.N; }
__device__ __forceinline__ int dev_tiles(const eagcn_batch& bt) { return min(bt.meta[EAGCN_META_NTILES], bt.n_tiles); }

// ---- column map of a layer's Fp-wide buffers -----------------------------------------------------
struct ViewCols {
    int K;
    int off[EAGCN_MAX_VIEWS + 1];   // padded column offsets, off[K] = Fp
    int width[EAGCN_MAX_VIEWS];     // exact widths
};
inline ViewCols view_cols(const eagcn_layer_params* p) {
    ViewCols v;
    v.K = p->K;
    int o = 0;
    for (int k = 0; k < p->K; ++k) {
        v.off[k] = o;
        v.width[k] = p->width[k];
        o += pad16(p->width[k]);
    }
    for (int k = p->K; k <= EAGCN_MAX_VIEWS; ++k) v.off[k] = o;
    for (int k = p->K; k < EAGCN_MAX_VIEWS; ++k) v.width[k] = 0;
    return v;
}
inline int layout_ld(const eagcn_layout* l) {
    int s = 0;
    for (int i = 0; i < l->nseg; ++i) s += l->pad[i];
    return s;
}
inline int layout_width(const eagcn_layout* l) {
    int s = 0;
    for (int i = 0; i < l->nseg; ++i) s += l->width[i];
    return s;
}

// ---- optional per-kernel-class timing (HIP events on the launch stream), off by default --------
enum ProfTag { PROF_INDEX = 0, PROF_PACK, PROF_GEMM, PROF_AGG, PROF_BN, PROF_EDGE, PROF_READOUT, PROF_HEAD, PROF_GEMM_PAIR,
               PROF_NTAGS };
bool prof_on();
void prof_begin(int tag, hipStream_t s, double work);
void prof_end(int tag, hipStream_t s);
struct ProfScope {
    int tag; hipStream_t s; bool on;
    ProfScope(int t, hipStream_t st, double work = 0.0) : tag(t), s(st), on(prof_on()) { if (on) prof_begin(tag, s, work); }
    ~ProfScope() { if (on) prof_end(tag, s); }
};

// ---- fork / join between the main stream and an auxiliary stream (events from a small pool; the
// record/wait pairs are legal inside stream capture and become graph dependencies) ---------------
hipEvent_t pool_event();
inline int stream_after(hipStream_t waiter, hipStream_t producer) {
    hipEvent_t e = pool_event();
    if (!e) return EAGCN_ERR_HIP;
    if (hipEventRecord(e, producer) != hipSuccess) return EAGCN_ERR_HIP;
    if (hipStreamWaitEvent(waiter, e, 0) != hipSuccess) return EAGCN_ERR_HIP;
    return EAGCN_OK;
}

// ---- internal launchers (defined across the .hip files) -----------------------------------------
struct GemmDesc {
    int ta, tb;            // 0: stored as used ([M][K] / [K][N]); 1: transposed storage
    int M, N, K;
    const float* A; int lda;
    const float* B; int ldb;
    float* C; int ldc;
    int splits;            // split-K: partial z written at C + z*slab
    size_t slab;           // floats between partial slabs
    double work = 0.0;     // algorithmic flops of this product (0 -> 2*M*N*K)
    int vecA = 1, vecB = 1; // set by launch_gemm: float4 loads allowed for A / B
    const int* M_dev = nullptr;   // if set: actual M (<= M) read on the device; M is then the capacity
    const int* K_dev = nullptr;   // if set: actual K (<= K), used by the split-K row reduction
    int prof_tag = PROF_GEMM;     // timing class (the head's small GEMMs are kept apart from the layer GEMMs)
};
int launch_gemm(const GemmDesc& g, hipStream_t s);
// dX (ta=0,tb=1) and dW (ta=1,tb=0) of one layer in one launch (falls back to two launches otherwise)
int launch_gemm_pair(const GemmDesc& dx, const GemmDesc& dw, hipStream_t s);

}  // namespace eagcn
