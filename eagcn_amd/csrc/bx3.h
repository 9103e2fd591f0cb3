// bf16 plane images of fp32 matrices: the operand format of csrc/gemm_bx3.hip, written by the kernels that PRODUCE a matrix.
//   three planes (exact):  x = x0 + x1 + x2, each piece the top 8 significand bits of what the pieces before it left over
//                          (every remainder is computed exactly in fp32), plane q at planes + q * pstride (elements);
//   one plane (bf16 mode): round to nearest even.
// Geometry of a plane: PANEL-MAJOR.  The columns are cut into panels of 32; a panel holds all `rows` (the row CAPACITY of the
// image) rows of its 32 columns, 64 bytes per row, and inside a row the four 16-byte chunks are stored XOR-swizzled by the row:
//     element (r, c)  ->  ((c >> 5) * rows + r) * 32  +  ((((c >> 3) & 3) ^ ((r >> 2) & 3)) << 3)  +  (c & 7)
// This IS the LDS image the GEMM multiplies from (gemm_bx3.hip): a 128-row x 32-k operand tile is 8 KB of CONTIGUOUS memory and
// a k-major 32-k x 128-column tile four contiguous 2 KB pieces, so every LDS-DMA instruction of the GEMM copies 1 KB of
// consecutive bytes, lane-linear.  (With row-major planes a k-contiguous tile was 128 segments of 64 bytes and the fill ran
// at 15-20 bytes per clock and CU against 23-56 for contiguous kilobytes: profiles/r04_lds_fill_probe.txt.)
#pragma once
#include "common.h"

namespace eagcn {

struct BxPlanes {                // bf16 planes of a matrix
    const uint16_t* p;           // plane 0; plane q at p + q * pstride
    size_t pstride;              // elements between planes (>= bx_plane_elems(rows, ld))
    int ld;                      // columns the image holds (a multiple of 8; the panels cover ceil(ld / 32) * 32)
    int rows;                    // row capacity of the image (the panel pitch)
};
struct BxOut {                   // the same, as an output of a producer kernel (p == nullptr: not requested)
    uint16_t* p; size_t pstride; int np; int rows;
};
// elements of one plane image
__host__ __device__ inline size_t bx_plane_elems(int rows, int ld) { return (size_t)((ld + 31) >> 5) * 32u * (size_t)(rows > 0 ? rows : 1); }
// element index of (r, c) inside a plane image of row capacity `rows`
__host__ __device__ inline size_t bx_addr(int rows, int r, int c) {
    return ((size_t)(c >> 5) * (size_t)rows + (size_t)r) * 32u + (size_t)(((((c >> 3) & 3) ^ ((r >> 2) & 3)) << 3) + (c & 7));
}

__device__ __forceinline__ uint32_t bx1_round(float v) {       // bf16 bits (in the HIGH half), round to nearest even
    const uint32_t b = __float_as_uint(v);
    return (b + 0x7FFFu + ((b >> 16) & 1u)) & 0xFFFF0000u;
}
// x = x0 + x1 + x2 exactly, every piece the bf16 NEAREST to what the pieces before it left over (remainders are exact in fp32:
// |x - x0| <= 2^-9 |x| has at most 16 significant bits, the next one at most 8).  Rounding instead of truncating makes the
// pieces' signs independent, so the three dropped piece products of a GEMM (gemm_bx3.hip) are <= 3 * 2^-26 |a b| and do not
// all pull towards zero (truncation measured a one-sided 2.5 eps of sum |a||b| that does not average out over k).
__device__ __forceinline__ void bx3_split(float v, uint32_t& h0, uint32_t& h1, uint32_t& h2) {
    h0 = bx1_round(v);
    const float r1 = v - __uint_as_float(h0);
    h1 = bx1_round(r1);
    const float r2 = r1 - __uint_as_float(h1);
    h2 = bx1_round(r2);
}
// four adjacent elements (idx a multiple of 4): one 8-byte store per plane
__device__ __forceinline__ void bx3_store4(uint16_t* __restrict__ pl, size_t pstride, size_t idx, const float4 v) {
    uint32_t h[3][4];
    bx3_split(v.x, h[0][0], h[1][0], h[2][0]);
    bx3_split(v.y, h[0][1], h[1][1], h[2][1]);
    bx3_split(v.z, h[0][2], h[1][2], h[2][2]);
    bx3_split(v.w, h[0][3], h[1][3], h[2][3]);
#pragma unroll
    for (int q = 0; q < 3; ++q)
        *reinterpret_cast<uint2*>(pl + (size_t)q * pstride + idx) = make_uint2((h[q][0] >> 16) | h[q][1], (h[q][2] >> 16) | h[q][3]);
}
__device__ __forceinline__ void bx1_store4(uint16_t* __restrict__ pl, size_t idx, const float4 v) {
    *reinterpret_cast<uint2*>(pl + idx) = make_uint2((bx1_round(v.x) >> 16) | bx1_round(v.y), (bx1_round(v.z) >> 16) | bx1_round(v.w));
}
// four adjacent columns c .. c + 3 (c a multiple of 4) of row r
__device__ __forceinline__ void bx_store4(const BxOut& o, int r, int c, const float4 v) {
    const size_t idx = bx_addr(o.rows, r, c);
    if (o.np == 3) bx3_store4(o.p, o.pstride, idx, v); else bx1_store4(o.p, idx, v);
}
// one element (scalar tails)
__device__ __forceinline__ void bx_store1(const BxOut& o, int r, int c, float v) {
    const size_t idx = bx_addr(o.rows, r, c);
    if (o.np == 3) {
        uint32_t h0, h1, h2;
        bx3_split(v, h0, h1, h2);
        o.p[idx] = (uint16_t)(h0 >> 16); o.p[o.pstride + idx] = (uint16_t)(h1 >> 16); o.p[2 * o.pstride + idx] = (uint16_t)(h2 >> 16);
    } else {
        o.p[idx] = (uint16_t)(bx1_round(v) >> 16);
    }
}

// one product of the plane GEMM (csrc/gemm_bx3.hip)
struct BxProb {
    BxPlanes A, B;
    float* C; int ldc;
    int M, N, K;                 // static extents (capacities where a device-side count exists)
    const int* M_dev;            // NT: actual rows of A / C
    const int* K_dev;            // TN: actual reduction length (packed rows)
    int tn;                      // 0: C = A[M,K] . B[N,K]^T;  1: C = A[K,M]^T . B[K,N]
    int splits; size_t slab;     // TN: k-chunk z goes to C + z * slab
};
// k-chunk slabs a split-K (TN) product of actual reduction length K over `tiles` output tiles really writes -- shared by the
// kernel and by whoever sums the slabs (layer.hip unpack_grads); the host sizes the slab buffer (`cap`) for a CAPACITY.
// Alone in its launch:
//   * enough chunks to give every CU about two work units, but none shorter than 24 k-tiles (768 rows: few enough slabs that
//     summing them stays cheap);
//   * never longer than 4096 rows (accumulation chains of gemm_bx3.hip: the bf16 MFMA accumulate drifts).
// Paired with an NT product of `ou` units of `okt` k-tiles each (the dX of the same layer; gemm_bx3.hip hands the units of both
// problems to the `per` workgroups of an XCD round-robin, the k-chunks first): the chunk count that minimises the LONGEST run of
// k-tiles any workgroup gets under that hand-out (chunks of at least 8 k-tiles, at most 4096 rows) -- at 4809 rows the fixed
// policy gave 5 of 32 workgroups a 25-k-tile chunk AND a 23-k-tile dX tile while 14 had 23.
// (cost of chunk count s under the hand-out: 8 x the longest run of k-tiles + s, so that near-ties take fewer slabs)
__host__ __device__ inline int bx3_split_cost(int s, int kt, int tiles, int n0x, int okt, int per) {
    const int b = n0x / per, q = n0x % per;
    const int c1 = (kt + s - 1) / s, n1x = (tiles * s + 7) >> 3, a = n1x / per, r = n1x % per;
    const int l1 = r > 0 ? (a + 1) * c1 + (b + (q > per - r ? 1 : 0)) * okt : 0;
    const int l2 = a * c1 + (b + (q > 0 ? 1 : 0)) * okt;
    return 8 * (l1 > l2 ? l1 : l2) + s;
}
__host__ __device__ inline void bx3_split_range(int cap, int K, int& kt, int& smin, int& smax) {
    kt = K > 32 ? (K + 31) / 32 : 1;
    const int drift = (K + 4095) / 4096;
    smin = drift > 1 ? drift : 1;
    smax = kt / 8;
    smin = smin < cap ? smin : cap;
    smax = smax < cap ? smax : cap;
    smax = smax > smin ? smax : smin;
    if (smax > smin + 63) smax = smin + 63;              // (64 candidates: one per lane of the wave-parallel form below)
}
__host__ __device__ inline int bx3_used_splits(int cap, int K, int tiles, int ou = 0, int okt = 0, int per = 32) {
    if (tiles < 1) tiles = 1;
    if (ou <= 0 || per <= 0) {
        const int kt = K > 32 ? (K + 31) / 32 : 1, drift = (K + 4095) / 4096;
        const int fill = (512 + tiles - 1) / tiles, len = kt / 24;
        int s = fill < len ? fill : len;
        s = s > drift ? s : drift;
        s = s < cap ? s : cap;
        return s > 1 ? s : 1;
    }
    int kt, smin, smax;
    bx3_split_range(cap, K, kt, smin, smax);
    int best = smin, best_cost = -1;
    for (int s = smin; s <= smax; ++s) {
        const int cost = bx3_split_cost(s, kt, tiles, (ou + 7) >> 3, okt, per);
        if (best_cost < 0 || cost < best_cost) { best_cost = cost; best = s; }
    }
    return best;
}
#if defined(__HIPCC__)
// the same choice, one candidate per lane (every lane of the calling wave must take part): the serial search above is up to 64 x
// three integer divisions -- evaluated by every thread of unpack_grads it made that launch 49 us instead of 14
__device__ __forceinline__ int bx3_used_splits_wave(int cap, int K, int tiles, int ou, int okt, int per) {
    if (ou <= 0 || per <= 0) return bx3_used_splits(cap, K, tiles);
    if (tiles < 1) tiles = 1;
    int kt, smin, smax;
    bx3_split_range(cap, K, kt, smin, smax);
    const int s = smin + (int)(threadIdx.x & 63);
    int key = s <= smax ? bx3_split_cost(s, kt, tiles, (ou + 7) >> 3, okt, per) * 1024 + s : 0x7FFFFFFF;      // (s < 1024: layer.hip caps the slab count)
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { const int other = __shfl_xor(key, o); key = other < key ? other : key; }
    return key & 1023;
}
#endif
// tile shapes: the 128 x 128 kernel (gemm_bx3.hip: 4 compute + 2 loader waves; launches of about one wave of tiles) and the 256 x 128
// kernel (gemm_bx3w.hip: 8 compute waves, two per SIMD; launches with several waves of tiles).  Whoever sums split-K slabs is told
// which one wrote them (the chunk policy counts tiles).
constexpr int BX3_BM = 128, BX3_BN = 128, BX3W_BM = 256, BX3W_BN = 128;
bool bx3_ok(const BxProb& p);
bool bx3_pair_policy();          // EAGCN_BX3_PAIR_POLICY=0: the chunks of a paired dW are sized as if it ran alone
int bx3_grid();
// np = 3: exact fp32 products from three planes; np = 1: plain bf16 operands.  p1 (optional): second product in the same launch
int launch_bx3(const BxProb& p0, const BxProb* p1, int np, hipStream_t s, double work, int prof_tag, int wide = 0);
int launch_bx3w(const BxProb& p0, const BxProb* p1, int np, hipStream_t s);          // gemm_bx3w.hip
// the host's choice between the two kernels for a launch whose NT problem has about `rows` actual rows (0: unknown -> capacity):
// EAGCN_BX3_WIDE = 0 never | 1 always | unset: by the number of 128 x 128 tiles against the CU count
int bx3_pick_wide(const BxProb& p0, const BxProb* p1, int rows_hint);
// fp32 [rows][ld] -> plane images of row capacity rows_cap
int launch_bx3_split(const float* x, int rows, const int* rows_dev, int ld, uint16_t* planes, size_t pstride, int rows_cap, int np, hipStream_t s);

}  // namespace eagcn
