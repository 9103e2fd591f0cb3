// Multi-view edge-attention aggregation on the matrix cores, and its backward pieces.
//
// Reference semantics (layers.py:82-92 with the masks of layers.py:294-304), per molecule b and
// view k, for the P_k = X.W_k slice produced by the flat GEMM (the reference computes (A.X).W,
// layers.py:39-40; the product is re-associated):
//     U[i,j]  = sigmoid(w_k[type(i,j)]) * adj[i,j] + sigmoid(self_r) * m_i * [i==j] + 1e-9*(1-adj[i,j])
//     A^[i,j] = m_i * U[i,j] / sum_j' U[i,j']          (j' over all N padded columns)
//     Y'[i,:] = sum_j A^[i,j] P_k[j,:]
// One wavefront owns one (molecule, view, 16-row tile): it walks the molecule's nat[b] columns in
// steps of 4 (the K extent of v_mfma_f32_16x16x4_f32), builds its A-operand value U[i, j0+q] on
// the fly from the uint8 bond-type map (one 16-byte load covers four k-steps) and an LDS sigma
// table, feeds the B operand straight from P (L1/L2 resident: a molecule's P slice is
// nat x F_k floats), and keeps CT 16x16 accumulators so each A value is reused CT times.  The row
// sum is accumulated in the same loop and applied once in the epilogue (row scaling commutes with
// the product), where the per-channel BatchNorm partial sums (sum y, sum y^2 in fp64) are taken as
// well.  No LDS tile, no barrier in the main loop, any molecule size.
// Small batches (B <= 256) use the K-split variant instead: one WORKGROUP per tile, the molecule's column
// groups spread over its four waves.  The CT column tiles are mapped so that a lane's four B-operand values /
// results form one float4 (256 contiguous bytes per row and 16 lanes).  BatchNorm partial sums live in LDS,
// the register budget is held at 128 (4 waves per SIMD): the kernels are bound by round trips in flight.
// In backward the transposed aggregation and the edge gradients of a layer share one grid (agg_edge_kernel).
// Columns j >= nat[b] hold 1e-9/rowsum weights on rows whose features are zero (Concate) or on a
// constant vector (Weighted_sum); they enter the row sum exactly and are dropped from the product
// (relative contribution <= N*1e-9).
#include <algorithm>
#include <type_traits>

#include <stdlib.h>

#include "common.h"
#include "kernels.h"

namespace eagcn {

// Variant for large batches: one WAVEFRONT per tile (four tiles per workgroup in flight); with thousands of
// small tiles this keeps 4x more tiles in flight than the K-split variant below.  Same operand layout as that variant: the CT
// column tiles are taken four at a time so that a lane's four B-operand values (and its four results) are ONE float4 -- 16
// adjacent lanes cover 256 contiguous bytes of a row (round 2 fed this kernel with 4-byte loads: four times the memory
// instructions of the K-split kernel for the same bytes, at the batch sizes where the aggregation is the largest item).
template <int CT, bool TRANS>
__device__ __forceinline__ void agg_wave_body(const AggArgs& a, const int bx, const int by, const int gx) {
    __shared__ float sig_s[256];
    __shared__ double st_s[TRANS ? 1 : CT * 16 * 2];
    const int k = by / a.nchunk, cc = by % a.nchunk;
    const int ntile_k = (a.vc.off[k + 1] - a.vc.off[k]) / 16;
    const int ct0 = cc * CT;
    if (ct0 >= ntile_k) return;                       // uniform for the whole workgroup
    const eagcn_batch& bt = a.bt;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // independent loads at the head of the wave, requested together: its first tile's descriptor (inside the tile CAPACITY:
    // always a legal address), the device-side tile count, sigmoid(self_r), the sigmoid table
    const int tile_id0 = bx * 4 + wave;               // the tile this wave takes first when tiles go out in dispatch order
    int4 ti_next = reinterpret_cast<const int4*>(bt.tile_info)[min(tile_id0, max(bt.n_tiles - 1, 0))];
    const float r = a.rsig[k];
    const float sig_v = a.sig[k * 256 + tid];
    const int ntiles = dev_tiles(a.bt);
    const int nlog = dev_n(a.bt);
    if (bx * 4 >= ntiles) return;        // capacity-sized grid: no tile for this workgroup (its
                                                      // stats slab is not read either: bn_finalize counts live slabs)
    // (the XCD-contiguous hand-out below only for batches of LARGE molecules -- four row tiles per molecule on average, from the
    //  device-side tile count: it makes the first descriptor a load that depends on that count, one more round trip per workgroup,
    //  and the slices of 19-atom molecules are a few rows that nobody shares)
    const bool xcd = a.xcd && ntiles >= 4 * bt.B;
    // Which tiles a workgroup takes.  Pass p of the grid covers the tiles [p gx 4, (p+1) gx 4); the tiles of a molecule are
    // consecutive and all of them read the molecule's slice of the source matrix.  Workgroup bx runs on XCD (bx + const) % 8:
    // handed out in dispatch order, the four workgroups that share a 256-atom molecule's slice sit on four XCDs and each of
    // the four L2s fetches it.  With a.xcd the workgroups of one XCD take a CONTIGUOUS range of the pass instead (a bijection
    // of the live workgroups of the pass, from the device-side tile count), so a slice is fetched into one L2.
    auto pass_tile = [&](int base) -> int {
        if (!xcd) return base + bx * 4 + wave;
        const int nl = min(gx, (ntiles - base + 3) >> 2);
        if (bx >= nl) return ntiles;
        const int q8 = nl >> 3, r8 = nl & 7, x = bx & 7;
        return base + ((x * q8 + min(x, r8) + (bx >> 3)) << 2) + wave;
    };
    const int tile_first = pass_tile(0);
    if (xcd) ti_next = reinterpret_cast<const int4*>(bt.tile_info)[min(tile_first, max(ntiles - 1, 0))];
    const int nct = min(CT, ntile_k - ct0);
    const int c0 = a.vc.off[k] + ct0 * 16;
    const int li = lane & 15, q = lane >> 4;
    constexpr int Q = CT / 4;
    const int ncol = nct * 16;                        // valid columns of this chunk
    auto tile_col0 = [&](int ct) { return ct < 4 * Q ? (ct >> 2) * 64 : Q * 64 + (ct - 4 * Q) * 16; };   // first column a tile touches
    auto lane_col = [&](int ct) { return ct < 4 * Q ? (ct >> 2) * 64 + 4 * li + (ct & 3) : Q * 64 + (ct - 4 * Q) * 16 + li; };
    // B operands of one source row for all tiles (zero where the column is outside the chunk)
    auto load_row = [&](const float* rowp, bool ok, float (&bv)[CT]) {
#pragma unroll
        for (int m = 0; m < Q; ++m) {
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (ok && m * 64 + 4 * li < ncol) v = *reinterpret_cast<const float4*>(rowp + m * 64 + 4 * li);
            bv[4 * m + 0] = v.x; bv[4 * m + 1] = v.y; bv[4 * m + 2] = v.z; bv[4 * m + 3] = v.w;
        }
#pragma unroll
        for (int ct = 4 * Q; ct < CT; ++ct) {
            const int c = Q * 64 + (ct - 4 * Q) * 16 + li;
            bv[ct] = (ok && c < ncol) ? rowp[c] : 0.0f;
        }
    };
    // the same for a row that is known to be inside and the first NA column tiles of the chunk (NA = CT, or 4 Q: the float4 groups
    // only -- an 8-tile layer runs the 9-tile instantiation with its last tile masked off): no predicate on any load
    auto load_row_full = [&](const float* rowp, float (&bv)[CT], auto na_c) {
        constexpr int NA = decltype(na_c)::value;
#pragma unroll
        for (int m = 0; m < Q; ++m) {
            const float4 v = *reinterpret_cast<const float4*>(rowp + m * 64 + 4 * li);
            bv[4 * m + 0] = v.x; bv[4 * m + 1] = v.y; bv[4 * m + 2] = v.z; bv[4 * m + 3] = v.w;
        }
#pragma unroll
        for (int ct = 4 * Q; ct < NA; ++ct) bv[ct] = rowp[Q * 64 + (ct - 4 * Q) * 16 + li];
    };
    sig_s[tid] = sig_v;
    if (!TRANS) for (int i = tid; i < CT * 16 * 2; i += 256) st_s[i] = 0.0;
    __syncthreads();
    // (BatchNorm partial sums are accumulated in LDS with fp64 atomics, not in 4*CT registers per lane: that keeps
    //  the kernel at four waves per SIMD)

    for (int base = 0, tile = tile_first; tile < ntiles; ) {
        const int4 ti = ti_next;
        base += gx * 4;
        tile = base < ntiles ? pass_tile(base) : ntiles;
        if (tile < ntiles) ti_next = reinterpret_cast<const int4*>(bt.tile_info)[tile];
        const int b = ti.x, rt = ti.y, n = ti.z, r0 = ti.w;
        const uint8_t* codeb = bt.code + ((size_t)k * bt.B + b) * bt.N * bt.ldc;
        const int ia = rt * 16 + li;                  // A-operand row of this lane = output row
        f32x4 acc[CT];
#pragma unroll
        for (int c = 0; c < CT; ++c) acc[c] = (f32x4){0.f, 0.f, 0.f, 0.f};
        const float* sbase = a.src + (size_t)r0 * a.lds + c0;

        if (!TRANS) {
            const float mi = (ia < n) ? bt.row_m[r0 + ia] : 0.0f;
            float dsum = 0.0f;
            int j0 = 0;
            // Blocks of 16 source rows that lie inside the molecule, for a row tile and a column chunk that are full: the same
            // arithmetic in the same order as the general block below with every predicate known -- no branch around any load
            // or MFMA (the general block compiles to ~60 branches per 16 columns and ran the matrix pipes at 41 % at N = 256:
            // profiles/r04_agg_sq_c5.txt).
            // Software pipeline over the blocks: the registers of k-step t are re-loaded for the NEXT block as soon as the MFMAs of
            // k-step t have read them, so every load has three k-steps of this wave's MFMAs (times the other waves' turns on the
            // pipe) to land in -- the kernel at N = 256 was a load round trip (~1.9 us under load) in front of every block, the
            // matrix pipes 40-47 % busy (profiles/r04_agg_sq_c5.txt).
            auto fast_fwd = [&](auto na_c) {
                constexpr int NA = decltype(na_c)::value;
                const uint8_t* crow = codeb + (size_t)ia * bt.ldc;
                if (j0 + 16 > n) return;
                float bv[4][CT];
                uint4 cw = *reinterpret_cast<const uint4*>(crow + j0);
#pragma unroll
                for (int t = 0; t < 4; ++t) load_row_full(sbase + (size_t)(j0 + 4 * t + q) * a.lds, bv[t], na_c);
                auto block = [&](auto more_c) {
                    constexpr bool MORE = decltype(more_c)::value;
                    const uint32_t w[4] = {cw.x, cw.y, cw.z, cw.w};
                    if constexpr (MORE) cw = *reinterpret_cast<const uint4*>(crow + j0 + 16);
                    float u4[4];
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        const uint32_t c = (w[t] >> (8 * q)) & 255u;
                        float u = sig_s[c] + (c == 0u ? TINY : 0.0f);
                        if (j0 + 4 * t + q == ia) u += r * mi;
                        dsum += u;
                        u4[t] = u;
                    }
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
#pragma unroll
                        for (int ct = 0; ct < NA; ++ct) acc[ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(u4[t], bv[t][ct], acc[ct], 0, 0, 0);
                        if constexpr (MORE) {
                            load_row_full(sbase + (size_t)(j0 + 16 + 4 * t + q) * a.lds, bv[t], na_c);
                            __builtin_amdgcn_sched_barrier(0);    // (the scheduler otherwise sinks all re-loads to the block's end)
                        }
                    }
                    j0 += 16;
                };
                while (j0 + 32 <= n) block(std::true_type{});
                block(std::false_type{});
            };
            if (rt * 16 + 16 <= n) {
                if (nct == CT) fast_fwd(std::integral_constant<int, CT>{});
                else if (Q > 0 && nct == 4 * Q) fast_fwd(std::integral_constant<int, (Q > 0 ? 4 * Q : CT)>{});
            }
            for (; j0 < n; j0 += 16) {
                uint4 cw = make_uint4(0u, 0u, 0u, 0u);
                if (ia < n) cw = *reinterpret_cast<const uint4*>(codeb + (size_t)ia * bt.ldc + j0);
                const uint32_t w[4] = {cw.x, cw.y, cw.z, cw.w};
                // all B-operand loads of the next four k-steps are issued before the first MFMA so
                // the L1/L2 latency is paid once per 16 columns, not once per k-step
                float bv[4][CT];
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const int j = j0 + 4 * t + q;
                    load_row(sbase + (size_t)min(j, n - 1) * a.lds, j < n, bv[t]);
                }
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const int jb = j0 + 4 * t;
                    if (jb < n) {
                        const int j = jb + q;
                        const uint32_t c = (w[t] >> (8 * q)) & 255u;
                        float u = sig_s[c] + (c == 0u ? TINY : 0.0f);
                        if (j == ia) u += r * mi;
                        if (j >= n || ia >= n) u = 0.0f;
                        dsum += u;
#pragma unroll
                        for (int ct = 0; ct < CT; ++ct)
                            if (tile_col0(ct) < ncol) acc[ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(u, bv[t][ct], acc[ct], 0, 0, 0);
                    }
                }
            }
            dsum += __shfl_xor(dsum, 16);
            dsum += __shfl_xor(dsum, 32);
            const float d = dsum + TINY * (float)(nlog - n);
            const float sc = (mi > 0.0f && ia < n) ? 1.0f / d : 0.0f;
            if (cc == 0 && q == 0 && ia < n) a.rscale[(size_t)k * bt.T + r0 + ia] = sc;
            float scr[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) scr[g] = __shfl(sc, q * 4 + g);
            // results scaled in place, BatchNorm partial sums per column (the four waves add concurrently), float4 / scalar stores
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) {
                const int lc = lane_col(ct);
                if (lc < ncol) {
                    double t1 = 0.0, t2 = 0.0;
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const int row = rt * 16 + q * 4 + g;
                        const float y = acc[ct][g] * scr[g];
                        acc[ct][g] = y;
                        if (row < n) { t1 += (double)y; t2 += (double)y * (double)y; }
                    }
                    t1 += __shfl_xor(t1, 16); t2 += __shfl_xor(t2, 16);
                    t1 += __shfl_xor(t1, 32); t2 += __shfl_xor(t2, 32);
                    if (q == 0) {
                        atomicAdd(&st_s[lc * 2 + 0], t1);
                        atomicAdd(&st_s[lc * 2 + 1], t2);
                    }
                }
            }
        } else {
            // dP[j,:] = sum_i A^[i,j] dY'[i,:]  (rscale carries m_i / rowsum_i)
            int i0 = 0;
            auto fast_trans = [&](auto na_c) {                // (see the forward form above)
                constexpr int NA = decltype(na_c)::value;
                const uint8_t* ccol = codeb + ia;
                const float* rsp = a.rscale + (size_t)k * bt.T + r0;
                if (i0 + 16 > n) return;
                uint32_t cc4[4];
                float rs4[4], bv[4][CT];
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const int i = i0 + 4 * t + q;
                    cc4[t] = ccol[(size_t)i * bt.ldc];
                    rs4[t] = rsp[i];
                    load_row_full(sbase + (size_t)i * a.lds, bv[t], na_c);
                }
                auto block = [&](auto more_c) {
                    constexpr bool MORE = decltype(more_c)::value;
                    float u4[4];
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        float u = sig_s[cc4[t]] + (cc4[t] == 0u ? TINY : 0.0f);
                        if (i0 + 4 * t + q == ia) u += r;
                        u4[t] = u * rs4[t];
                        if constexpr (MORE) {
                            const int i = i0 + 16 + 4 * t + q;
                            cc4[t] = ccol[(size_t)i * bt.ldc];
                            rs4[t] = rsp[i];
                        }
                    }
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
#pragma unroll
                        for (int ct = 0; ct < NA; ++ct) acc[ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(u4[t], bv[t][ct], acc[ct], 0, 0, 0);
                        if constexpr (MORE) {
                            load_row_full(sbase + (size_t)(i0 + 16 + 4 * t + q) * a.lds, bv[t], na_c);
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    }
                    i0 += 16;
                };
                while (i0 + 32 <= n) block(std::true_type{});
                block(std::false_type{});
            };
            if (rt * 16 + 16 <= n) {
                if (nct == CT) fast_trans(std::integral_constant<int, CT>{});
                else if (Q > 0 && nct == 4 * Q) fast_trans(std::integral_constant<int, (Q > 0 ? 4 * Q : CT)>{});
            }
            for (; i0 < n; i0 += 16) {
                uint32_t cc4[4];
                float rs4[4], bv[4][CT];
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const int i = i0 + 4 * t + q;
                    cc4[t] = 0u;
                    rs4[t] = 0.0f;
                    if (i < n && ia < n) {
                        cc4[t] = codeb[(size_t)i * bt.ldc + ia];
                        rs4[t] = a.rscale[(size_t)k * bt.T + r0 + i];
                    }
                    load_row(sbase + (size_t)min(i, n - 1) * a.lds, i < n, bv[t]);
                }
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    if (i0 + 4 * t < n) {
                        const int i = i0 + 4 * t + q;
                        float u = sig_s[cc4[t]] + (cc4[t] == 0u ? TINY : 0.0f);
                        if (i == ia) u += r;
                        u *= rs4[t];
#pragma unroll
                        for (int ct = 0; ct < CT; ++ct)
                            if (tile_col0(ct) < ncol) acc[ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(u, bv[t][ct], acc[ct], 0, 0, 0);
                    }
                }
            }
        }
        // stores: a lane's four results of a row are one float4 (columns 4 li .. 4 li + 3 of a 64-column block)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int row = rt * 16 + q * 4 + g;
            if (row < n) {
                if (TRANS && a.planes.p) {            // dP for the plane GEMMs (gemm_bx3.hip): split here, once
#pragma unroll
                    for (int m = 0; m < Q; ++m)
                        if (m * 64 + 4 * li < ncol)
                            bx_store4(a.planes, r0 + row, c0 + m * 64 + 4 * li, make_float4(acc[4 * m][g], acc[4 * m + 1][g], acc[4 * m + 2][g], acc[4 * m + 3][g]));
#pragma unroll
                    for (int ct = 4 * Q; ct < CT; ++ct)
                        if (lane_col(ct) < ncol) bx_store1(a.planes, r0 + row, c0 + lane_col(ct), acc[ct][g]);
                    continue;
                }
                float* drow = a.dst + (size_t)(r0 + row) * a.ldd + c0;
#pragma unroll
                for (int m = 0; m < Q; ++m)
                    if (m * 64 + 4 * li < ncol)
                        *reinterpret_cast<float4*>(drow + m * 64 + 4 * li) =
                            make_float4(acc[4 * m][g], acc[4 * m + 1][g], acc[4 * m + 2][g], acc[4 * m + 3][g]);
#pragma unroll
                for (int ct = 4 * Q; ct < CT; ++ct)
                    if (lane_col(ct) < ncol) drow[lane_col(ct)] = acc[ct][g];
            }
        }
    }

    if (!TRANS) {
        // per-workgroup partial BatchNorm sums (accumulated in st_s) -> slab[bx][column][2]
        __syncthreads();
        const int fp = a.vc.off[a.vc.K];
        for (int i = tid; i < nct * 16 * 2; i += 256)
            a.stats[((size_t)bx * fp + c0) * 2 + i] = st_s[i];
    }
}

// One WORKGROUP owns one (molecule, view, 16-row tile) and its four wavefronts split the K range: wave w
// takes the 16-column groups g = w, w+4, ...  The kernel's duration is set by the largest molecule of the
// batch (a 132-atom molecule = 9 dependent column groups in one wave, while a typical 18-atom tile is
// done after 2); splitting the groups over the waves cuts that chain 4x.  Partial accumulators and partial
// row sums are combined through LDS by wave 0, which also runs the epilogue.
template <int CT, bool TRANS, int NW = 4>
__device__ __forceinline__ void agg_body(const AggArgs& a, const int bx, const int by, const int gx) {
    __shared__ float sig_s[256];
    __shared__ float red_s[CT][4][64];                // ONE partial-accumulator buffer [tile][register][lane], reused wave by wave
    __shared__ float dred_s[4][16];
    __shared__ double st_s[TRANS ? 1 : CT * 16 * 2];
    const int k = by / a.nchunk, cc = by % a.nchunk;
    const int ntile_k = (a.vc.off[k + 1] - a.vc.off[k]) / 16;
    const int ct0 = cc * CT;
    if (ct0 >= ntile_k) return;                       // uniform for the whole workgroup
    const eagcn_batch& bt = a.bt;
    // Independent loads at the head of every workgroup are requested together instead of one dependent round
    // trip after the other: the first tile's descriptor (bx is inside the tile CAPACITY, so the address is
    // valid even when this workgroup turns out to have no tile), the device-side tile count, sigmoid(self_r)
    // and the sigmoid table.
    int4 ti_next = reinterpret_cast<const int4*>(bt.tile_info)[bx];
    const float r = a.rsig[k];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float sig_v[4 / NW];                              // NW*64 threads fetch the 256-entry table
#pragma unroll
    for (int i = 0; i < 4 / NW; ++i) sig_v[i] = a.sig[k * 256 + tid + i * NW * 64];
    const int ntiles = dev_tiles(a.bt);
    const int nlog = dev_n(a.bt);
    if (bx >= ntiles) return;            // capacity-sized grid: no tile for this workgroup (its
                                                      // stats slab is not read either: bn_finalize counts live slabs)
    const int nct = min(CT, ntile_k - ct0);
    const int c0 = a.vc.off[k] + ct0 * 16;
    const int li = lane & 15, q = lane >> 4;
    // Column map of the CT 16-wide MFMA tiles: tiles are taken four at a time so that a lane's four B-operand
    // values (and its four results) are ONE float4 = 16 adjacent lanes cover 256 contiguous bytes of a row
    // instead of four 64-byte pieces (half the memory requests, whole 128-byte lines):
    //   ct <  4Q: column = 64*(ct/4) + 4*li + ct%4         ct >= 4Q: column = 64*Q + 16*(ct-4Q) + li
    constexpr int Q = CT / 4;
    const int ncol = nct * 16;                        // valid columns of this chunk
    auto tile_col0 = [&](int ct) { return ct < 4 * Q ? (ct >> 2) * 64 : Q * 64 + (ct - 4 * Q) * 16; };   // first column a tile touches
    auto lane_col = [&](int ct) { return ct < 4 * Q ? (ct >> 2) * 64 + 4 * li + (ct & 3) : Q * 64 + (ct - 4 * Q) * 16 + li; };
    // B operands of one source row for all tiles (zero where the column is outside the chunk)
    auto load_row = [&](const float* rowp, bool ok, float (&bv)[CT]) {
#pragma unroll
        for (int m = 0; m < Q; ++m) {
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (ok && m * 64 + 4 * li < ncol) v = *reinterpret_cast<const float4*>(rowp + m * 64 + 4 * li);
            bv[4 * m + 0] = v.x; bv[4 * m + 1] = v.y; bv[4 * m + 2] = v.z; bv[4 * m + 3] = v.w;
        }
#pragma unroll
        for (int ct = 4 * Q; ct < CT; ++ct) {
            const int c = Q * 64 + (ct - 4 * Q) * 16 + li;
            bv[ct] = (ok && c < ncol) ? rowp[c] : 0.0f;
        }
    };
#pragma unroll
    for (int i = 0; i < 4 / NW; ++i) sig_s[tid + i * NW * 64] = sig_v[i];
    if (!TRANS) for (int i = tid; i < CT * 16 * 2; i += NW * 64) st_s[i] = 0.0;
    __syncthreads();
    // (BatchNorm partial sums live in LDS, st_s, not in registers: 36 VGPRs less for CT = 9, i.e. one more
    //  resident workgroup per CU for a kernel whose speed is the number of round trips in flight)

    for (int tile = bx; tile < ntiles; tile += gx) {
        const int4 ti = ti_next;
        if (tile + gx < ntiles) ti_next = reinterpret_cast<const int4*>(bt.tile_info)[tile + gx];
        const int b = ti.x, rt = ti.y, n = ti.z, r0 = ti.w;
        const uint8_t* codeb = bt.code + ((size_t)k * bt.B + b) * bt.N * bt.ldc;
        const int ia = rt * 16 + li;                  // A-operand row of this lane = output row
        const int ngroups = (n + 15) >> 4;
        const int nw = min(NW, ngroups);               // waves that have a share of the K range
        f32x4 acc[CT];
#pragma unroll
        for (int c = 0; c < CT; ++c) acc[c] = (f32x4){0.f, 0.f, 0.f, 0.f};
        const float* sbase = a.src + (size_t)r0 * a.lds + c0;
        float dsum = 0.0f;

        if (!TRANS) {
            const float mi = (ia < n) ? bt.row_m[r0 + ia] : 0.0f;
            for (int grp = wave; grp < ngroups; grp += NW) {
                const int j0 = grp * 16;
                uint4 cw = make_uint4(0u, 0u, 0u, 0u);
                if (ia < n) cw = *reinterpret_cast<const uint4*>(codeb + (size_t)ia * bt.ldc + j0);
                const uint32_t w[4] = {cw.x, cw.y, cw.z, cw.w};
                // all B-operand loads of the four k-steps are issued before the first MFMA
                float bv[4][CT];
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const int j = j0 + 4 * t + q;
                    load_row(sbase + (size_t)min(j, n - 1) * a.lds, j < n, bv[t]);
                }
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const int jb = j0 + 4 * t;
                    if (jb < n) {
                        const int j = jb + q;
                        const uint32_t c = (w[t] >> (8 * q)) & 255u;
                        float u = sig_s[c] + (c == 0u ? TINY : 0.0f);
                        if (j == ia) u += r * mi;
                        if (j >= n || ia >= n) u = 0.0f;
                        dsum += u;
#pragma unroll
                        for (int ct = 0; ct < CT; ++ct)
                            if (tile_col0(ct) < ncol) acc[ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(u, bv[t][ct], acc[ct], 0, 0, 0);
                    }
                }
            }
            dsum += __shfl_xor(dsum, 16);
            dsum += __shfl_xor(dsum, 32);
        } else {
            // dP[j,:] = sum_i A^[i,j] dY'[i,:]  (rscale carries m_i / rowsum_i)
            for (int grp = wave; grp < ngroups; grp += NW) {
                const int i0 = grp * 16;
                uint32_t cc4[4];
                float rs4[4], bv[4][CT];
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const int i = i0 + 4 * t + q;
                    cc4[t] = 0u;
                    rs4[t] = 0.0f;
                    if (i < n && ia < n) {
                        cc4[t] = codeb[(size_t)i * bt.ldc + ia];
                        rs4[t] = a.rscale[(size_t)k * bt.T + r0 + i];
                    }
                    load_row(sbase + (size_t)min(i, n - 1) * a.lds, i < n, bv[t]);
                }
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    if (i0 + 4 * t < n) {
                        const int i = i0 + 4 * t + q;
                        float u = sig_s[cc4[t]] + (cc4[t] == 0u ? TINY : 0.0f);
                        if (i == ia) u += r;
                        u *= rs4[t];
#pragma unroll
                        for (int ct = 0; ct < CT; ++ct)
                            if (tile_col0(ct) < ncol) acc[ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(u, bv[t][ct], acc[ct], 0, 0, 0);
                    }
                }
            }
        }

        // combine the waves' partials in wave 0: waves 1..nw-1 hand their accumulators over one after the other
        // through a single LDS buffer (9 KB for CT = 9; three parallel buffers would cap residency at 5 WG/CU)
        if (nw > 1) {
            if (!TRANS && q == 0 && wave < nw) dred_s[wave][li] = dsum;
            for (int w2 = 1; w2 < nw; ++w2) {
                if (wave == w2) {
#pragma unroll
                    for (int ct = 0; ct < CT; ++ct)
                        if (tile_col0(ct) < ncol) {
#pragma unroll
                            for (int g = 0; g < 4; ++g) red_s[ct][g][lane] = acc[ct][g];
                        }
                }
                __syncthreads();
                if (wave == 0) {
#pragma unroll
                    for (int ct = 0; ct < CT; ++ct)
                        if (tile_col0(ct) < ncol) {
#pragma unroll
                            for (int g = 0; g < 4; ++g) acc[ct][g] += red_s[ct][g][lane];
                        }
                    if (!TRANS) dsum += dred_s[w2][li];
                }
                if (w2 + 1 < nw) __syncthreads();
            }
        }
        if (wave == 0) {
            if (!TRANS) {
                const float mi = (ia < n) ? bt.row_m[r0 + ia] : 0.0f;
                const float d = dsum + TINY * (float)(nlog - n);
                const float sc = (mi > 0.0f && ia < n) ? 1.0f / d : 0.0f;
                if (cc == 0 && q == 0 && ia < n) a.rscale[(size_t)k * bt.T + r0 + ia] = sc;
                float scr[4];
#pragma unroll
                for (int g = 0; g < 4; ++g) scr[g] = __shfl(sc, q * 4 + g);
                // results scaled in place, BatchNorm partial sums per column, then float4 / scalar stores
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) {
                    const int lc = lane_col(ct);
                    if (lc < ncol) {
                        double t1 = 0.0, t2 = 0.0;
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            const int row = rt * 16 + q * 4 + g;
                            const float y = acc[ct][g] * scr[g];
                            acc[ct][g] = y;
                            if (row < n) { t1 += (double)y; t2 += (double)y * (double)y; }
                        }
                        t1 += __shfl_xor(t1, 16); t2 += __shfl_xor(t2, 16);
                        t1 += __shfl_xor(t1, 32); t2 += __shfl_xor(t2, 32);
                        if (q == 0) {                 // only wave 0 touches st_s between the barriers
                            st_s[lc * 2 + 0] += t1;
                            st_s[lc * 2 + 1] += t2;
                        }
                    }
                }
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int row = rt * 16 + q * 4 + g;
                    if (row < n) {
                        float* drow = a.dst + (size_t)(r0 + row) * a.ldd + c0;
#pragma unroll
                        for (int m = 0; m < Q; ++m)
                            if (m * 64 + 4 * li < ncol)
                                *reinterpret_cast<float4*>(drow + m * 64 + 4 * li) =
                                    make_float4(acc[4 * m][g], acc[4 * m + 1][g], acc[4 * m + 2][g], acc[4 * m + 3][g]);
#pragma unroll
                        for (int ct = 4 * Q; ct < CT; ++ct)
                            if (lane_col(ct) < ncol) drow[lane_col(ct)] = acc[ct][g];
                    }
                }
            } else {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int row = rt * 16 + q * 4 + g;
                    if (row < n) {
                        if (a.planes.p) {             // dP for the plane GEMMs (gemm_bx3.hip): split here, once
#pragma unroll
                            for (int m = 0; m < Q; ++m)
                                if (m * 64 + 4 * li < ncol)
                                    bx_store4(a.planes, r0 + row, c0 + m * 64 + 4 * li, make_float4(acc[4 * m][g], acc[4 * m + 1][g], acc[4 * m + 2][g], acc[4 * m + 3][g]));
#pragma unroll
                            for (int ct = 4 * Q; ct < CT; ++ct)
                                if (lane_col(ct) < ncol) bx_store1(a.planes, r0 + row, c0 + lane_col(ct), acc[ct][g]);
                            continue;
                        }
                        float* drow = a.dst + (size_t)(r0 + row) * a.ldd + c0;
#pragma unroll
                        for (int m = 0; m < Q; ++m)
                            if (m * 64 + 4 * li < ncol)
                                *reinterpret_cast<float4*>(drow + m * 64 + 4 * li) =
                                    make_float4(acc[4 * m][g], acc[4 * m + 1][g], acc[4 * m + 2][g], acc[4 * m + 3][g]);
#pragma unroll
                        for (int ct = 4 * Q; ct < CT; ++ct)
                            if (lane_col(ct) < ncol) drow[lane_col(ct)] = acc[ct][g];
                    }
                }
            }
        }
        if (nw > 1) __syncthreads();                  // red_s / dred_s are reused by the next tile
    }

    if (!TRANS) {
        // per-workgroup partial BatchNorm sums (accumulated in st_s by wave 0) -> slab[bx][column][2]
        __syncthreads();
        const int fp = a.vc.off[a.vc.K];
        for (int i = tid; i < nct * 16 * 2; i += NW * 64)
            a.stats[((size_t)bx * fp + c0) * 2 + i] = st_s[i];
    }
}

// (register budget: 128 = four waves per SIMD up to six column tiles; the software pipeline of agg_wave_body needs more from seven tiles
//  on -- at 128 the 9-tile instantiations that BASELINE's wide layers run spilled inside their MFMA loops -- and gets
//  168 = three waves per SIMD there; EAGCN_AGG_WIDE_REGS in tools/r5_agg_regs.sh is the A/B)
template <int CT, bool TRANS>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(CT >= 7 ? 3 : 4, 8))) void agg_wave_kernel(AggArgs a) { agg_wave_body<CT, TRANS>(a, blockIdx.x, blockIdx.y, gridDim.x); }
template <int CT, bool TRANS>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 8))) void agg_kernel(AggArgs a) { agg_body<CT, TRANS>(a, blockIdx.x, blockIdx.y, gridDim.x); }
// two waves per workgroup: most tiles have one or two column groups, so two of four waves would idle
template <int CT, bool TRANS>
__global__ __launch_bounds__(128) __attribute__((amdgpu_waves_per_eu(4, 8))) void agg2_kernel(AggArgs a) { agg_body<CT, TRANS, 2>(a, blockIdx.x, blockIdx.y, gridDim.x); }

template <bool TRANS>
static int launch_agg_t(const AggArgs& a, int ct, dim3 grid, bool ksplit, hipStream_t s) {
    switch (ct) {
    static const int nw2 = [] { const char* e = getenv("EAGCN_AGG_NW"); return e ? atoi(e) : 4; }();
#define EAGCN_AGG_CASE(N) case N: if (ksplit && nw2 == 2) agg2_kernel<N, TRANS><<<grid, 128, 0, s>>>(a); \
                                  else if (ksplit) agg_kernel<N, TRANS><<<grid, 256, 0, s>>>(a); \
                                  else agg_wave_kernel<N, TRANS><<<grid, 256, 0, s>>>(a); break;
        EAGCN_AGG_CASE(1) EAGCN_AGG_CASE(2) EAGCN_AGG_CASE(3) EAGCN_AGG_CASE(4) EAGCN_AGG_CASE(5)
        EAGCN_AGG_CASE(6) EAGCN_AGG_CASE(7) EAGCN_AGG_CASE(8) EAGCN_AGG_CASE(9) EAGCN_AGG_CASE(10)
#undef EAGCN_AGG_CASE
        default: set_error("agg: unsupported CT %d", ct); return EAGCN_ERR_ARG;
    }
    EAGCN_LAUNCH_CHECK();
    return EAGCN_OK;
}

// column tiles per workgroup: at most 9 (register budget: the 10-tile variants spill a few registers in every form, and the
// 8-tile forward variant of the wave-per-tile kernel spills 492 bytes per lane -- two full float4 groups and no scalar tile make
// the allocator give up; an 8-tile layer runs the 9-tile instantiation with its last tile masked off instead);
// EAGCN_AGG_MAXCT changes the cap (1..10)
static int agg_max_ct() {
    static const int v = [] { const char* e = getenv("EAGCN_AGG_MAXCT"); const int x = e ? atoi(e) : 9; return x < 1 ? 1 : (x > 10 ? 10 : x); }();
    return v;
}
// wave-per-tile variant: XCD-contiguous tile handout (agg_wave_body); EAGCN_AGG_XCD=0 hands the tiles out in dispatch order
static int agg_xcd() {
    static const int v = [] { const char* e = getenv("EAGCN_AGG_XCD"); return e ? atoi(e) : 1; }();
    return v;
}
static int agg_pick_ct(int tmax, int nchunk) {
    const int ct = cdiv(tmax, nchunk);
    return ct == 8 ? 9 : ct;
}
// number of workgroups along x (= number of stat partial slabs) used for a batch
// Small batches (few hundred tiles, duration set by the largest molecule): one workgroup per tile, K split
// over its waves.  Large batches: one wave per tile.  Measured on MI355X in round 1: K-split 115 vs 136 us/step at
// B=256, but 377 vs 253 us/step at B=1024; round 4 (16-byte operand loads, branch-free pipelined block loop in the
// wave-per-tile kernel): equal at B = 256 (0.455 ms either way), wave-per-tile ahead at B = 512 (Tox21 0.641 -> 0.616 ms,
// Lipo 3-layer 1.557 -> 1.439 ms): the switch is at 256 (EAGCN_AGG_KSPLIT_MAXB).
bool agg_ksplit(const eagcn_batch* b) {
    static const int maxb = [] { const char* e = getenv("EAGCN_AGG_KSPLIT_MAXB"); return e ? atoi(e) : 256; }();
    return b->B <= maxb;
}
// number of workgroups along x = number of BatchNorm stat slabs
int agg_grid_x(const eagcn_batch* b) {
    return agg_ksplit(b) ? std::max(1, std::min(b->n_tiles, 1024)) : std::max(1, std::min(cdiv(b->n_tiles, 4), 512));
}

int launch_agg(AggArgs a, bool trans, hipStream_t s) {
    if (a.bt.n_tiles == 0) return EAGCN_OK;
    int tmax = 0;
    for (int k = 0; k < a.vc.K; ++k) tmax = std::max(tmax, (a.vc.off[k + 1] - a.vc.off[k]) / 16);
    // balanced chunking: fewest chunks of at most 10 column tiles, then the smallest CT reaching it
    const int nchunk = cdiv(tmax, agg_max_ct());
    const int ct = agg_pick_ct(tmax, nchunk);
    a.nchunk = nchunk;
    a.xcd = agg_xcd();
    dim3 grid(agg_grid_x(&a.bt), a.vc.K * nchunk);
    ProfScope ps(PROF_AGG, s);
    const bool ks = agg_ksplit(&a.bt);
    return trans ? launch_agg_t<true>(a, ct, grid, ks, s) : launch_agg_t<false>(a, ct, grid, ks, s);
}

// ---- edge gradients ------------------------------------------------------------------------------
// Backward of the attention build (closed form in SURVEY.md 8a):
//   dA^[i,j]  = <dY'[i,:], P[j,:]>            needed only where U depends on a parameter
//   rowdot_i  = sum_l dA^[i,l] A^[i,l] = <dY'[i,:], Y'[i,:]>
//   dU[i,j]   = (m_i / rowsum_i) (dA^[i,j] - rowdot_i)
//   d w_k[c] += dU[i,j] s (1-s)  at bonds of type c ;   d self_r_k += dU[i,i] r (1-r)
// one wavefront per (packed row, view); results accumulated in fp64.
__device__ __forceinline__ void edge_grad_body(const EdgeArgs& a, const int bx, const int by, const int gx) {
    // one packed row per 16-lane group (4 rows per wavefront, 16 per workgroup).
    // Dependent-load hops per row: {row_info, rscale} -> {code row, dY row, Y row} -> {P rows of up to four
    // bonds at once}.  The code row arrives with ONE 16-byte load per lane (16 lanes x 16 bytes = 256 columns);
    // the bonds found in it are queued in LDS and their dot products are taken four at a time.
    constexpr int HMAX = 64;                                       // queue slots per row (flushed when full)
    __shared__ float sig_s[256];
    __shared__ double h_s[256];
    __shared__ double dr_s[16];
    __shared__ int hit_s[16][HMAX];                                // (column << 8) | bond code
    const eagcn_batch& bt = a.bt;
    const int k = by;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int grp = lane >> 4, sl = lane & 15;
    const int gq = wave * 4 + grp;                                 // group index inside the workgroup
    // first row's descriptor and scale, the sigmoid table and the device-side row count are independent:
    // request them together (the row index is inside the row CAPACITY, so the loads are always legal)
    const int r_first = (bx * 4 + wave) * 4 + grp;
    int4 info_first = make_int4(0, 0, 0, 0);
    float rs_first = 0.0f;
    if (r_first < bt.T) {
        info_first = reinterpret_cast<const int4*>(bt.row_info)[r_first];
        rs_first = a.rscale[(size_t)k * bt.T + r_first];
    }
    const float sig_v = a.sig[k * 256 + tid];
    const int Tn = dev_rows(bt);
    if (bx * 16 >= Tn) return;                        // capacity-sized grid (slab not read either)
    const int nwg = min(gx, (Tn + 15) / 16);           // workgroups that have rows
    sig_s[tid] = sig_v;
    h_s[tid] = 0.0;
    if (tid < 16) dr_s[tid] = 0.0;
    __syncthreads();
    const int off = a.vc.off[k], fp = a.vc.off[k + 1] - a.vc.off[k];
    double dr_acc = 0.0;
    // Which row blocks a workgroup takes.  A row's bonds point at rows of the same molecule: the P rows it gathers are its
    // molecule's slice.  Handed out in dispatch order the 16 workgroups of a 256-atom molecule sit on all eight XCDs; with a.xcd
    // the workgroups of one XCD (same bx mod 8) take a contiguous range of the live blocks (a bijection, as in agg_wave_body) and
    // the gathers of a molecule meet in one L2.  Measured at K = 8 / N = 256: agg_edge<9> 2.94 -> 2.73 ms, agg_edge<4> 1.87 -> 1.63 ms,
    // the step 16.1 -> 15.4 ms -- with the memory-side FETCH_SIZE of the launch unchanged (5.06e6 KiB raw): what was saved is L2
    // miss latency on the gather chains, not fabric bytes.
    // (only for batches of LARGE molecules -- 64 packed rows per molecule on average, from the device-side row count: the remapped
    //  block cannot use the descriptor prefetched above, one more dependent round trip per workgroup, which costs a batch of
    //  19-atom molecules 6 % of its step and buys it nothing: their slices are a few rows)
    int blk0 = bx;
    if (a.xcd && Tn >= 64 * bt.B) {
        const int q8 = nwg >> 3, r8 = nwg & 7, x = bx & 7;
        blk0 = x * q8 + min(x, r8) + (bx >> 3);
    }
    for (int rblk = blk0; rblk * 16 < Tn; rblk += nwg) {     // one trip unless the grid was capped
        const int r = (rblk * 4 + wave) * 4 + grp;
        int4 info = make_int4(0, 0, 0, 0);
        float rs = 0.0f;
        if (r < Tn) {
            if (rblk == bx) { info = info_first; rs = rs_first; }
            else {
                info = reinterpret_cast<const int4*>(bt.row_info)[r];
                rs = a.rscale[(size_t)k * bt.T + r];
            }
        }
        const bool live = rs != 0.0f;                              // m_i == 0 rows carry no dependence
        const int b = info.x, i = info.y, n = live ? info.z : 0, r0 = info.w;
        const int rr = live ? r : 0;
        const float* dy = a.dY + (size_t)rr * a.ld + off;
        const float* yr = a.Y + (size_t)rr * a.ld + off;
        const uint8_t* crow = bt.code + (((size_t)k * bt.B + b) * bt.N + i) * bt.ldc;
        const int nmax = max(max(__shfl(n, 0), __shfl(n, 16)), max(__shfl(n, 32), __shfl(n, 48)));
        // (view segments are zero-padded to a multiple of 16 columns and 16-byte aligned: 16-byte loads whenever the row stride is)
        const bool vec = (a.ld & 3) == 0 && ((reinterpret_cast<uintptr_t>(a.dY) | reinterpret_cast<uintptr_t>(a.Y) | reinterpret_cast<uintptr_t>(a.P)) & 15) == 0 && (off & 3) == 0;
        float rd = !live ? 0.0f : (vec ? dot16v(dy, yr, sl, fp) : dot16(dy, yr, sl, fp));
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) rd += __shfl_xor(rd, o);
        int nh = 0;                                                // queued hits of this group (group-uniform)
        auto flush = [&]() {                                       // dot products of the queued bonds, 4 at a time
            // the queue was written by other lanes of this wavefront: LDS operations of one wave complete in
            // order; the fence keeps the compiler from moving the reads above the writes
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            for (int h0 = 0; h0 < nh; h0 += 4) {
                int hv[4];
                const float* pr[4];
#pragma unroll
                for (int h = 0; h < 4; ++h) {
                    hv[h] = (h0 + h < nh) ? hit_s[gq][h0 + h] : -1;
                    pr[h] = a.P + (size_t)(hv[h] >= 0 ? (r0 + (hv[h] >> 8)) : rr) * a.ld + off;
                }
                float g[4];
                if (vec) dot16x4v(dy, pr, sl, fp, g); else dot16x4(dy, pr, sl, fp, g);
#pragma unroll
                for (int h = 0; h < 4; ++h) {
#pragma unroll
                    for (int o = 8; o > 0; o >>= 1) g[h] += __shfl_xor(g[h], o);
                    if (hv[h] >= 0 && sl == 0) {
                        const float dU = rs * (g[h] - rd);
                        const uint32_t cj = (uint32_t)hv[h] & 255u;
                        if (cj) {
                            const float sg = sig_s[cj];
                            atomicAdd(&h_s[cj], (double)(dU * sg * (1.0f - sg)));
                        }
                        if ((hv[h] >> 8) == i) dr_acc += (double)dU;
                    }
                }
            }
            nh = 0;
        };
        for (int seg = 0; seg * 256 < nmax; ++seg) {
            const int jb = seg * 256 + sl * 16;
            uint4 cw = make_uint4(0u, 0u, 0u, 0u);
            if (jb < n) cw = *reinterpret_cast<const uint4*>(crow + jb);
            const uint32_t w[4] = {cw.x, cw.y, cw.z, cw.w};
#pragma unroll
            for (int t = 0; t < 16; ++t) {
                const uint32_t c = (w[t >> 2] >> (8 * (t & 3))) & 255u;
                const int j = jb + t;
                const bool want = j < n && (c != 0u || j == i);
                const unsigned long long ball = __ballot(want);
                if (ball == 0ull) continue;                            // wave-uniform
                const uint32_t mask = (uint32_t)(ball >> (grp * 16)) & 0xFFFFu;   // this group's lanes with a hit at byte t
                if (mask) {
                    // lane `sl` with a hit takes slot nh + (number of hit lanes below it)
                    const int pos = nh + __popc(mask & ((1u << sl) - 1u));
                    if (want && pos < HMAX) hit_s[gq][pos] = (j << 8) | (int)c;
                    nh = min(nh + __popc(mask), HMAX);
                }
                if (nh > HMAX - 16) flush();                           // keep room for the next byte position
            }
        }
        flush();
    }
    if (sl == 0) dr_s[wave * 4 + grp] = dr_acc;
    __syncthreads();
    // slab[bx][k][0..255] = bond-type histogram, slab[..][k][256] = self term -- or the non-zero bins added to one of the shared
    // accumulator slabs
    if (a.atomic) {
        double* out = a.datt + ((size_t)(bx & (EDGE_COPIES - 1)) * a.vc.K + k) * EDGE_SLAB;
        const double v = h_s[tid];
        if (v != 0.0) atomicAdd(&out[tid], v);
        if (tid == 0) {
            double t = 0.0;
            for (int q = 0; q < 16; ++q) t += dr_s[q];
            if (t != 0.0) atomicAdd(&out[256], t);
        }
    } else {
        double* out = a.datt + ((size_t)bx * a.vc.K + k) * EDGE_SLAB;
        out[tid] = h_s[tid];
        if (tid == 0) {
            double t = 0.0;
            for (int q = 0; q < 16; ++q) t += dr_s[q];
            out[256] = t;
        }
    }
}

__global__ __launch_bounds__(256) void edge_grad_kernel(EdgeArgs a) { edge_grad_body(a, blockIdx.x, blockIdx.y, gridDim.x); }

// Transposed aggregation and edge gradients of one layer in ONE grid: both only read dY' (and saved
// activations), both are chains of dependent loads that leave most of a CU idle, and neither fills the chip
// alone -- side by side they overlap instead of running back to back.  Workgroups [0, agg_gx) of every grid row
// run the aggregation, the rest the edge gradients (grid rows beyond the view count have none).
template <int CT, bool KSPLIT>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu((CT >= 7 && !KSPLIT) ? 3 : 4, 8))) void agg_edge_kernel(AggArgs a, EdgeArgs e, int agg_gx) {
    const int bx = blockIdx.x;
    if (bx < agg_gx) {
        if constexpr (KSPLIT) agg_body<CT, true>(a, bx, blockIdx.y, agg_gx);
        else agg_wave_body<CT, true>(a, bx, blockIdx.y, agg_gx);
    } else if ((int)blockIdx.y < e.vc.K) {
        edge_grad_body(e, bx - agg_gx, blockIdx.y, (int)gridDim.x - agg_gx);
    }
}

// one workgroup per 16 packed rows; workgroups beyond the actual row count exit at once
int edge_grid_x(const eagcn_batch* b) { return std::max(1, std::min(cdiv(b->T, 16), 1024)); }

int launch_edge_grad(const EdgeArgs& a_in, hipStream_t s) {
    EdgeArgs a = a_in;
    a.xcd = (agg_xcd() && !agg_ksplit(&a.bt)) ? 1 : 0;
    if (a.bt.T == 0) return EAGCN_OK;
    dim3 grid(edge_grid_x(&a.bt), a.vc.K);
    ProfScope ps(PROF_EDGE, s);
    edge_grad_kernel<<<grid, 256, 0, s>>>(a);
    EAGCN_LAUNCH_CHECK();
    return EAGCN_OK;
}

// transposed aggregation + edge gradients in one launch (see agg_edge_kernel)
int launch_agg_edge(AggArgs a, const EdgeArgs& e_in, hipStream_t s) {
    EdgeArgs e = e_in;
    e.xcd = (agg_xcd() && !agg_ksplit(&a.bt)) ? 1 : 0;
    if (a.bt.n_tiles == 0 || a.bt.T == 0) return EAGCN_OK;
    int tmax = 0;
    for (int k = 0; k < a.vc.K; ++k) tmax = std::max(tmax, (a.vc.off[k + 1] - a.vc.off[k]) / 16);
    const int nchunk = cdiv(tmax, agg_max_ct());
    const int ct = agg_pick_ct(tmax, nchunk);
    a.nchunk = nchunk;
    a.xcd = agg_xcd();
    const int agx = agg_grid_x(&a.bt), egx = edge_grid_x(&a.bt);
    dim3 grid(agx + egx, a.vc.K * nchunk);
    const bool ks = agg_ksplit(&a.bt);
    ProfScope ps(PROF_AGG, s);
    switch (ct) {
#define EAGCN_AE_CASE(N) case N: if (ks) agg_edge_kernel<N, true><<<grid, 256, 0, s>>>(a, e, agx); \
                                 else agg_edge_kernel<N, false><<<grid, 256, 0, s>>>(a, e, agx); break;
        EAGCN_AE_CASE(1) EAGCN_AE_CASE(2) EAGCN_AE_CASE(3) EAGCN_AE_CASE(4) EAGCN_AE_CASE(5)
        EAGCN_AE_CASE(6) EAGCN_AE_CASE(7) EAGCN_AE_CASE(8) EAGCN_AE_CASE(9) EAGCN_AE_CASE(10)
#undef EAGCN_AE_CASE
        default: set_error("agg: unsupported CT %d", ct); return EAGCN_ERR_ARG;
    }
    EAGCN_LAUNCH_CHECK();
    return EAGCN_OK;
}

}  // namespace eagcn
