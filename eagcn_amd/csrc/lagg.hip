// Edge-attention aggregation over bond lists with the operand STAGED IN LDS: the operator of agg.hip / sagg.hip (reference
// layers.py:82-92 with the masks of layers.py:294-304), per molecule b and view k
//     U[i,j]  = sigmoid(w_k[type(i,j)]) adj[i,j] + sigmoid(self_r) m_i [i == j] + 1e-9 (1 - adj[i,j])
//     A^[i,j] = m_i U[i,j] / sum_j' U[i,j'] ,   Y'[i,:] = sum_j A^[i,j] P_k[j,:]
// evaluated as what it is -- sigma at the two to four bonds of an atom, sigma(self_r) on the diagonal, the constant 1e-9 elsewhere:
//     sum_j U[i,j] P[j,:] = sum_{bonds (i,j)} sigma_ij P[j,:] + r m_i P[i,:] + 1e-9 ( S_b - sum_{bonds (i,j)} P[j,:] ),   S_b = sum_{j < nat} P[j,:]
// a gather of deg + 1 rows plus one rank-one term per molecule (exact, filler included: sagg.hip).
//
// Why a third form.  The matrix-core kernels (agg.hip) multiply the full nat x nat block: at 256 atoms
// 99 % of their MFMAs multiply by 1e-9, and over small molecules with wide layers (HIV: 1250 columns per view) they re-read every
// operand row once per 16-row tile and stream at 1-1.7 TB/s.  The bond-list kernel of round 4 (sagg.hip) did 64x fewer multiply-adds
// and lost anyway: a wavefront owned a molecule and gathered its neighbour rows from L2, every batch of rows one memory round trip
// behind the previous one.  Here the round trips are taken ONCE per workgroup, for everything, and the gathers are LDS reads:
//   * a workgroup owns a ROW BLOCK -- whole molecules, greedily packed to <= 256 packed rows / 16 molecules (eagcn_batch.blk, built
//     with the batch index) -- and a 32-column chunk of one view;
//   * it loads the block's rows of the operand (one 16-byte load per lane and row: 128 contiguous bytes per row), the rows' list
//     headers and the block's list entries (contiguous: the entries of molecule b live in [edge0[b], edge0[b+1])) in ONE batch of
//     independent loads, and puts them into LDS (32 KB of operand; list entries as {atom, sigma});
//   * S_b: 32 groups of 8 lanes sum contiguous row ranges and add them to the molecule's LDS slot;
//   * then 8 lanes own a row: deg + 1 `ds_read_b128` gathers, a handful of FMAs, one 16-byte store.  BatchNorm partial sums (fp64)
//     are carried per lane and leave as one slab per row block (layer.hip bn_finalize counts the live blocks from meta[NBLK]).
// Every operand element is read from memory once and every result written once; the matrix cores are not involved (there is no
// dense block to multiply).
// Backward (one kernel): the transposed aggregation dP[j,:] = sum_{bonds (i,j)} s_i sigma_ij dY'[i,:] + s_j r dY'[j,:] + 1e-9 (G_b -
// sum_{bonds} s_i dY'[i,:]), G_b = sum_i s_i dY'[i,:], s_i = m_i / rowsum_i -- and, from the SAME gathers, the edge gradients of
// agg.hip's edge_grad_kernel: with dY' of the block in LDS, row j's own P and Y' rows in registers,
//     dA^[i,j] = <dY'[i,:], P[j,:]> ,  rowdot_i = <dY'[i,:], Y'[i,:]> ,  dU[i,j] = s_i (dA^[i,j] - rowdot_i)
// are partial sums over the workgroup's 32 columns (everything is linear in them): d w_k[c] += dU s (1 - s) by bond type through an
// LDS histogram, d self_r from the diagonal, flushed with fp64 atomics into the shared accumulator slabs (kernels.h EDGE_COPIES).
#include <stdlib.h>

#include <algorithm>

#include "common.h"
#include "kernels.h"

namespace eagcn {

constexpr int LG_CW = 32;                    // columns per workgroup
constexpr int LG_LPR = LG_CW / 4;            // lanes per row (16 bytes each)
constexpr int LG_G = 256 / LG_LPR;           // row groups per workgroup (32)
constexpr int LG_U = LAGG_RB / LG_G;         // staging loads per lane (8)
constexpr int LG_ECAP = 1024;                // list entries of a block staged in LDS (the rest is read from memory)
constexpr int LG_EPT = LG_ECAP / 256;

__device__ __forceinline__ void lg_fma(float4& acc, float w, const float4& v) {
    acc.x = fmaf(w, v.x, acc.x); acc.y = fmaf(w, v.y, acc.y); acc.z = fmaf(w, v.z, acc.z); acc.w = fmaf(w, v.w, acc.w);
}
__device__ __forceinline__ void lg_add(float4& acc, const float4& v) { acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w; }
__device__ __forceinline__ float lg_dot(const float4& a, const float4& b) { return fmaf(a.x, b.x, fmaf(a.y, b.y, fmaf(a.z, b.z, a.w * b.w))); }
// sum over the LG_LPR lanes of a row group (the groups are aligned 8-lane runs of a wavefront)
__device__ __forceinline__ float lg_gsum(float v) {
    v += __shfl_xor(v, 1); v += __shfl_xor(v, 2); v += __shfl_xor(v, 4);
    return v;
}

template <bool TRANS>
__global__ __launch_bounds__(256) void lagg_kernel(AggArgs a, EdgeArgs ed) {
    __shared__ float4 buf[LAGG_RB][LG_LPR];          // the block's operand rows x this chunk's columns (32 KB)
    __shared__ int2 s_ptr[LAGG_RB];                  // per row: {first list entry relative to the block, count}
    __shared__ float s_rs[LAGG_RB];                  // forward: m_i; transposed: s_i = m_i / rowsum_i
    __shared__ float s_rd[TRANS ? LAGG_RB : 1];      // transposed: this chunk's part of rowdot_i = <dY'_i, Y'_i>
    __shared__ unsigned char s_rm[LAGG_RB];          // molecule of the row (index inside the block)
    __shared__ float4 s_S[LAGG_MAXM][LG_LPR];        // S_b / G_b per molecule
    __shared__ unsigned short s_nb[LG_ECAP];         // list entry: atom inside its molecule
    __shared__ float s_w[LG_ECAP];                   //             sigma of the bond
    __shared__ unsigned char s_cd[TRANS ? LG_ECAP : 1];   //        bond-type code (edge gradients)
    __shared__ float sig_s[256];
    __shared__ unsigned short s_mo[LAGG_RB];         // first row of the row's molecule inside the block
    __shared__ double st_s[TRANS ? 1 : 4][TRANS ? 1 : LG_LPR][8];   // forward: per wave: BatchNorm partial sums of a lane's four columns
    __shared__ double h_s[TRANS ? 264 : 1];          // transposed: bond-type histogram of d w_k, [256] = d self_r

    const eagcn_batch& bt = a.bt;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int k = blockIdx.x / a.nchunk, cc = blockIdx.x - k * a.nchunk;
    const int wk = a.vc.off[k + 1] - a.vc.off[k];                    // padded width of the view (a multiple of 16)
    if (cc * LG_CW >= wk) return;                                     // (uniform)
    const int nblk = bt.meta[EAGCN_META_NBLK];
    const int nlog = dev_n(bt);
    const float r = a.rsig[k];
    sig_s[tid] = a.sig[k * 256 + tid];                               // (made visible by the first barrier of the block loop)
    const int l = tid & (LG_LPR - 1), g = tid / LG_LPR;
    const int col = cc * LG_CW + 4 * l;                               // this lane's first column inside the view
    const bool col_ok = col < wk;
    const int c0 = a.vc.off[k] + col;
    const float* rsk = a.rscale + (size_t)k * bt.T;
    const int2* ptrs = reinterpret_cast<const int2*>(TRANS ? bt.col_ptr : bt.row_ptr);
    const int32_t* nbr = TRANS ? bt.tnbr : bt.nbr;
    const uint64_t* codes = TRANS ? bt.tcode : bt.ecode;
    if constexpr (TRANS) { h_s[tid] = 0.0; if (tid < 8) h_s[256 + tid] = 0.0; }
    double dr_acc = 0.0;
    // (the grid's y extent is an estimate of the block count: a workgroup takes blocks q, q + gridDim.y, ...)
    const int dbg = a.xcd;                                            // (probe mask, EAGCN_LAGG_DBG: wrong results)
    const int4* blk4 = reinterpret_cast<const int4*>(bt.blk);
    const int4* rinfo = reinterpret_cast<const int4*>(bt.row_info);
    // A workgroup takes the blocks q, q + gridDim.y, ... (the grid's y extent is an estimate of the block count).  Measured and
    // dropped: a software pipeline over a workgroup's blocks (the next block's rows in flight into registers while this one is worked
    // on, a persistent grid of three workgroups per CU): 212 / 238 registers = two workgroups per CU instead of three, and slower --
    // 613 / 1242 us forward / backward against 420 / 792 at the HIV widths, 1151 / 1954 against 1040 / 1229 at 256 atoms: the phases
    // between the barriers are chains of dependent LDS reads that only MORE resident workgroups hide.
    const int c0s = col_ok ? c0 : a.vc.off[k];                        // (a legal column for the lanes beyond the view's width)
    for (int q = blockIdx.y; q < nblk; q += gridDim.y) {
    // the block: {first molecule, molecules, first packed row, rows} {first list entry, entries} -- one dependent load, then everything
    const int4 b0 = blk4[2 * q], b1 = blk4[2 * q + 1];
    const int m0 = b0.x, R0 = b0.z, rows = min(b0.w, LAGG_RB), E0 = b1.x, ne = b1.y;
    if (rows <= 0) {                                                  // (uniform) nothing stored: the slab still has to be defined
        if constexpr (!TRANS) {
            const int fp = a.vc.off[a.vc.K];
            if (tid < LG_CW && cc * LG_CW + tid < wk)
                *reinterpret_cast<double2*>(a.stats + ((size_t)q * fp + a.vc.off[k] + cc * LG_CW + tid) * 2) = make_double2(0.0, 0.0);
        }
        continue;
    }
    // ---- ONE batch of independent loads: operand rows, (transposed) the rows' own Y', row descriptors, list headers, list entries.
    //      Every load is unconditional on a clamped address (a load under a per-lane condition compiles to a branch and, behind it, a
    //      wait per load); what a lane must not use is zeroed afterwards.
    float4 v[LG_U], yv[TRANS ? LG_U : 1];
#pragma unroll
    for (int u = 0; u < LG_U; ++u) {
        const int rc = min(g + LG_G * u, rows - 1);
        v[u] = *reinterpret_cast<const float4*>(a.src + (size_t)(R0 + rc) * a.lds + c0s);
        if constexpr (TRANS) yv[u] = *reinterpret_cast<const float4*>(ed.Y + (size_t)(R0 + rc) * ed.ld + c0s);
    }
    const int tr = R0 + min(tid, rows - 1);
    const int2 pt = ptrs[tr];
    const float rsv = TRANS ? rsk[tr] : bt.row_m[tr];
    const int4 ri = rinfo[tr];                                        // {molecule, atom, nat, first row of the molecule}
    int e_jn[LG_EPT];
    uint64_t e_cd[LG_EPT];
    const int nst = min(ne, LG_ECAP);
    if (nst > 0) {                                                    // (uniform)
#pragma unroll
        for (int u = 0; u < LG_EPT; ++u) {
            const int ec = E0 + min(tid + 256 * u, nst - 1);
            e_jn[u] = nbr[ec];
            e_cd[u] = codes[ec];
        }
    }
    __syncthreads();                                                  // (LDS of the previous block is free)
    if (tid < LAGG_MAXM * LG_LPR) (&s_S[0][0])[tid] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int u = 0; u < LG_U; ++u) {
        const int rr = g + LG_G * u;
        if (!col_ok) { v[u] = make_float4(0.f, 0.f, 0.f, 0.f); if constexpr (TRANS) yv[u] = make_float4(0.f, 0.f, 0.f, 0.f); }
        if (rr < rows) buf[rr][l] = v[u];
        if constexpr (TRANS) {
            // this chunk's part of rowdot_i (the operands are in registers)
            const float d = lg_gsum(lg_dot(v[u], yv[u]));
            if (rr < rows && l == 0) s_rd[rr] = d;
        }
    }
    if (tid < rows) {
        s_ptr[tid] = make_int2(pt.x - E0, pt.y);
        s_rs[tid] = rsv;
        s_rm[tid] = (unsigned char)min(max(ri.x - m0, 0), LAGG_MAXM - 1);
        s_mo[tid] = (unsigned short)(ri.w - R0);
    }
    if (nst > 0) {
#pragma unroll
        for (int u = 0; u < LG_EPT; ++u) {
            const int e = tid + 256 * u;
            if (e < nst) {
                const uint32_t c = (uint32_t)(e_cd[u] >> (8 * k)) & 255u;
                s_nb[e] = (unsigned short)e_jn[u];
                s_w[e] = sig_s[c];
                if constexpr (TRANS) s_cd[e] = (unsigned char)c;
            }
        }
    }
    __syncthreads();                                                  // B2
    // ---- S_b (forward) / G_b = sum_i s_i dY'_i (transposed): group g sums a contiguous range of the block's rows -------------------
    if (!(dbg & 1)) {
        const int per = (rows + LG_G - 1) / LG_G;
        const int ra = g * per, rb = min(rows, ra + per);
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        int cur = -1;
        auto flush = [&]() {
            if (cur >= 0) {
                float* dst = reinterpret_cast<float*>(&s_S[cur][l]);
                atomicAdd(dst + 0, acc.x); atomicAdd(dst + 1, acc.y); atomicAdd(dst + 2, acc.z); atomicAdd(dst + 3, acc.w);
            }
        };
        for (int rr = ra; rr < rb; ++rr) {
            const int m = s_rm[rr];
            if (m != cur) { flush(); acc = make_float4(0.f, 0.f, 0.f, 0.f); cur = m; }
            if constexpr (TRANS) lg_fma(acc, s_rs[rr], buf[rr][l]); else lg_add(acc, buf[rr][l]);
        }
        flush();
    }
    __syncthreads();                                                  // B3
    // entry `el` of the block's lists: {atom inside its molecule, sigma, code}
    auto entry = [&](int el, int& jn, float& w, uint32_t& c) __attribute__((always_inline)) {
        if (el < LG_ECAP) {
            jn = s_nb[el]; w = s_w[el];
            if constexpr (TRANS) c = s_cd[el]; else c = 0u;
        } else {                                                      // (beyond the staged part: memory)
            jn = nbr[E0 + el];
            c = (uint32_t)(codes[E0 + el] >> (8 * k)) & 255u;
            w = sig_s[c];
        }
    };
    double s1[4] = {0.0, 0.0, 0.0, 0.0}, s2[4] = {0.0, 0.0, 0.0, 0.0};
    // ---- the rows: 8 lanes own a row ---------------------------------------------------------------------------------------------------
    for (int rr = g; rr < rows; rr += LG_G) {
        int2 p = s_ptr[rr];
        if (dbg & 2) p.y = 0;                                         // (probe: no gathers)
        const int m = s_rm[rr];
        const int moff = s_mo[rr];                                    // first row of the molecule inside the block
        const float mrow = s_rs[rr];                                  // forward: m_i; transposed: s_j
        const float4 self = buf[rr][l];
        const float4 S = s_S[m][l];
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f), bs = make_float4(0.f, 0.f, 0.f, 0.f);
        float wsum = 0.0f;
        float4 pj = make_float4(0.f, 0.f, 0.f, 0.f);
        if constexpr (TRANS) {
            pj = *reinterpret_cast<const float4*>(ed.P + (size_t)(R0 + rr) * ed.ld + c0s);
            if (!col_ok) pj = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        for (int e0 = 0; e0 < p.y; e0 += 4) {
            int jn[4]; float ww[4]; uint32_t cd[4]; float4 vv[4]; float sw[4];
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                jn[d] = 0; ww[d] = 0.0f; cd[d] = 0u; sw[d] = 0.0f;
                if (e0 + d < p.y) entry(p.x + e0 + d, jn[d], ww[d], cd[d]);
                const int src_row = min(moff + jn[d], LAGG_RB - 1);
                vv[d] = buf[src_row][l];
                if constexpr (TRANS) sw[d] = (e0 + d < p.y) ? s_rs[src_row] : 0.0f;
            }
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                if constexpr (!TRANS) {
                    lg_fma(acc, ww[d], vv[d]);
                    if (e0 + d < p.y) lg_add(bs, vv[d]);
                    wsum += ww[d];
                } else {
                    lg_fma(acc, ww[d] * sw[d], vv[d]);
                    lg_fma(bs, sw[d], vv[d]);                         // (what the 1e-9 term must NOT count: s_i dY'_i of the bonded rows)
                    // edge gradient of bond (i -> j): dU = s_i (<dY'_i, P_j> - rowdot_i), this chunk's share
                    const float gd = lg_gsum(lg_dot(vv[d], pj));
                    if (e0 + d < p.y && l == 0 && sw[d] != 0.0f) {
                        const int src_row = min(moff + jn[d], LAGG_RB - 1);
                        const float dU = sw[d] * (gd - s_rd[src_row]);
                        if (cd[d]) atomicAdd(&h_s[cd[d]], (double)(dU * ww[d] * (1.0f - ww[d])));
                        if (src_row == rr) dr_acc += (double)dU;      // (a self bond: the diagonal term below is NOT taken again)
                    }
                }
            }
        }
        const int row = R0 + rr;
        float4 y;
        if constexpr (!TRANS) {
            // rowsum_i = sum sigma + r m_i + 1e-9 (columns without a bond, padding included)
            const float d = wsum + r * mrow + TINY * (float)(nlog - p.y);
            const float sc = mrow > 0.0f ? 1.0f / d : 0.0f;
            if (cc == 0 && l == 0) a.rscale[(size_t)k * bt.T + row] = sc;
            const float rm = r * mrow;
            y.x = sc * (acc.x + rm * self.x + TINY * (S.x - bs.x));
            y.y = sc * (acc.y + rm * self.y + TINY * (S.y - bs.y));
            y.z = sc * (acc.z + rm * self.z + TINY * (S.z - bs.z));
            y.w = sc * (acc.w + rm * self.w + TINY * (S.w - bs.w));
            s1[0] += (double)y.x; s2[0] += (double)y.x * (double)y.x;
            s1[1] += (double)y.y; s2[1] += (double)y.y * (double)y.y;
            s1[2] += (double)y.z; s2[2] += (double)y.z * (double)y.z;
            s1[3] += (double)y.w; s2[3] += (double)y.w * (double)y.w;
        } else {
            const float rs = r * mrow;                                // s_j r
            y.x = acc.x + rs * self.x + TINY * (S.x - bs.x);
            y.y = acc.y + rs * self.y + TINY * (S.y - bs.y);
            y.z = acc.z + rs * self.z + TINY * (S.z - bs.z);
            y.w = acc.w + rs * self.w + TINY * (S.w - bs.w);
            // the diagonal of the edge gradients (always part of d self_r; agg.hip edge_grad_body)
            const float gd = lg_gsum(lg_dot(self, pj));
            if (l == 0 && mrow != 0.0f) {
                bool self_bond = false;                               // (already counted above when the list holds (j, j))
                for (int e = 0; e < p.y; ++e) { int jn; float w; uint32_t c; entry(p.x + e, jn, w, c); self_bond = self_bond || (moff + jn == rr); }
                if (!self_bond) dr_acc += (double)(mrow * (gd - s_rd[rr]));
            }
        }
        if (col_ok && !(dbg & 4)) {
            if (TRANS && a.planes.p) bx_store4(a.planes, row, c0, y);
            else *reinterpret_cast<float4*>(a.dst + (size_t)row * a.ldd + c0) = y;
        }
    }
    if constexpr (!TRANS) {
        // BatchNorm partial sums: the eight groups of a wave hold the same columns -> wave sum, then the four waves through LDS
#pragma unroll
        for (int e = 0; e < 4; ++e) {
#pragma unroll
            for (int o = LG_LPR; o < 64; o <<= 1) { s1[e] += __shfl_xor(s1[e], o); s2[e] += __shfl_xor(s2[e], o); }
        }
        if (lane < LG_LPR) {
#pragma unroll
            for (int e = 0; e < 4; ++e) { st_s[wave][lane][2 * e] = s1[e]; st_s[wave][lane][2 * e + 1] = s2[e]; }
        }
        __syncthreads();
        if (tid < LG_CW && cc * LG_CW + tid < wk) {
            const int fp = a.vc.off[a.vc.K];
            const int ll = tid >> 2, e = tid & 3;
            const double t1 = (st_s[0][ll][2 * e] + st_s[1][ll][2 * e]) + (st_s[2][ll][2 * e] + st_s[3][ll][2 * e]);
            const double t2 = (st_s[0][ll][2 * e + 1] + st_s[1][ll][2 * e + 1]) + (st_s[2][ll][2 * e + 1] + st_s[3][ll][2 * e + 1]);
            *reinterpret_cast<double2*>(a.stats + ((size_t)q * fp + a.vc.off[k] + cc * LG_CW + tid) * 2) = make_double2(t1, t2);
        }
    }
    }                                                                 // (blocks)
    if constexpr (TRANS) {
        if (l == 0 && dr_acc != 0.0) atomicAdd(&h_s[256], dr_acc);
        __syncthreads();
        // non-zero bins -> one of the shared accumulator slabs (kernels.h EDGE_COPIES; drained by unpack_grads)
        double* out = ed.datt + ((size_t)((blockIdx.y + blockIdx.x) & (EDGE_COPIES - 1)) * ed.vc.K + k) * EDGE_SLAB;
        const double hv = h_s[tid];
        if (hv != 0.0) atomicAdd(&out[tid], hv);
        if (tid == 0 && h_s[256] != 0.0) atomicAdd(&out[256], h_s[256]);
    }
}

// ---- policy ------------------------------------------------------------------------------------------------------------------------------
// EAGCN_AGG = lds (always, N <= 256) | dense | sparse (sagg.hip); default: by shape.  Measured on MI355X (profiles/r05_lagg_*; us per launch,
// forward / backward with the edge gradients, against agg_wave + agg_edge of agg.hip):
//     K = 8, N = 256, all molecules 256 atoms, B = 1024 (BASELINE configs[4]):   1040 / 1230  against  1340 / 2180   (step 14.6 -> 12.5 ms)
//     HIV widths (5 x 1250 columns), molecules of 24 atoms on average, N = 222:     420 /  790  against   380 /  840   (equal)
//     Tox21 batch 1024 (19 atoms on average, N = 132):                                52 /  114  against    38 /   90   (slower)
//     Tox21 batch 256:                                                                21 /   38  against    21 /   35   (equal; its bond
//                                                   lists cost the side stream 0.2 ms per batch that a 0.45 ms step cannot hide)
// i.e. it pays where the dense block is mostly filler: LARGE molecules.  The lists are built (and the path taken) for padded sizes from
// EAGCN_LAGG_MIN_N (240) and, once the rows batches of the shape really hold are known (eagcn_batch.t_hint), 96 atoms per molecule.
static int lagg_policy() {
    static const int v = [] {
        const char* e = getenv("EAGCN_AGG");
        if (e && !strcmp(e, "lds")) return 1;
        if (e && (!strcmp(e, "dense") || !strcmp(e, "sparse"))) return 0;
        return 2;                                     // by shape
    }();
    return v;
}
static int lagg_min_n() {
    static const int v = [] { const char* e = getenv("EAGCN_LAGG_MIN_N"); return e ? atoi(e) : 240; }();
    return v;
}
bool lagg_wanted(int B, int N) {
    const int p = lagg_policy();
    if (p == 0 || N > LAGG_RB) return false;
    if (p == 1) return true;
    return N >= lagg_min_n();
}
int lagg_parts() {                                    // (debug switch) bit 0: forward, bit 1: backward on this path
    static const int v = [] { const char* e = getenv("EAGCN_LAGG_PARTS"); return e ? atoi(e) : 3; }();
    return v;
}
bool lagg_use(const eagcn_batch* b) {
    if (!(b->build_lists && b->blk && b->mol_info && b->row_ptr && b->col_ptr && b->nbr && b->tnbr && b->ecode && b->tcode && b->row_info &&
          lagg_wanted(b->B, b->N))) return false;
    if (lagg_policy() == 2 && b->t_hint > 0 && (long)b->t_hint < 96L * b->B) return false;       // small molecules in a large padding
    return true;
}
int lagg_slabs(const eagcn_batch* b) { return std::max(1, b->B); }

static int lagg_grid(const AggArgs& a, dim3* grid, int* nchunk) {
    int wmax = 0;
    for (int k = 0; k < a.vc.K; ++k) wmax = std::max(wmax, a.vc.off[k + 1] - a.vc.off[k]);
    *nchunk = cdiv(wmax, LG_CW);
    // y: an ESTIMATE of the block count from the rows batches of this shape hold (two consecutive blocks together exceed LAGG_RB rows
    // or LAGG_MAXM molecules); the kernel loops, so any count is handled, and a tight grid spares the launch thousands of workgroups
    // that would only find out that they have no block
    const int rows = a.bt.t_hint > 0 ? std::min(a.bt.t_hint, a.bt.T) : a.bt.T;
    const int est = 2 * (rows / LAGG_RB + a.bt.B / LAGG_MAXM) + 2;
    *grid = dim3((unsigned)(a.vc.K * *nchunk), (unsigned)std::max(1, std::min(std::min(a.bt.B, est), 65535)));
    return EAGCN_OK;
}

int launch_lagg_fwd(AggArgs a, hipStream_t s) {
    if (a.bt.B == 0 || a.bt.T == 0) return EAGCN_OK;
    dim3 grid;
    int rc = lagg_grid(a, &grid, &a.nchunk);
    if (rc) return rc;
    ProfScope ps(PROF_AGG, s);
    EdgeArgs e;
    memset(&e, 0, sizeof(e));
    static const int dbgm = [] { const char* e = getenv("EAGCN_LAGG_DBG"); return e ? atoi(e) : 0; }();
    a.xcd = dbgm;
    lagg_kernel<false><<<grid, 256, 0, s>>>(a, e);
    EAGCN_LAUNCH_CHECK();
    return EAGCN_OK;
}

int launch_lagg_bwd(AggArgs a, const EdgeArgs& e, hipStream_t s) {
    if (a.bt.B == 0 || a.bt.T == 0) return EAGCN_OK;
    EAGCN_CHECK_ARG(e.atomic && e.datt, "lagg: the edge gradients leave through the shared accumulator slabs (EdgeArgs.atomic)");
    dim3 grid;
    int rc = lagg_grid(a, &grid, &a.nchunk);
    if (rc) return rc;
    ProfScope ps(PROF_AGG, s);
    static const int dbgm = [] { const char* e = getenv("EAGCN_LAGG_DBG"); return e ? atoi(e) : 0; }();
    a.xcd = dbgm;
    lagg_kernel<true><<<grid, 256, 0, s>>>(a, e);
    EAGCN_LAUNCH_CHECK();
    return EAGCN_OK;
}

}  // namespace eagcn
