// Edge-attention aggregation over bond lists with the operand STAGED IN LDS: the operator of agg.hip (reference
// layers.py:82-92 with the masks of layers.py:294-304), per molecule b and view k
//     U[i,j]  = sigmoid(w_k[type(i,j)]) adj[i,j] + sigmoid(self_r) m_i [i == j] + 1e-9 (1 - adj[i,j])
//     A^[i,j] = m_i U[i,j] / sum_j' U[i,j'] ,   Y'[i,:] = sum_j A^[i,j] P_k[j,:]
// evaluated as what it is -- sigma at the two to four bonds of an atom, sigma(self_r) on the diagonal, the constant 1e-9 elsewhere:
//     sum_j U[i,j] P[j,:] = sum_{bonds (i,j)} sigma_ij P[j,:] + r m_i P[i,:] + 1e-9 ( S_b - sum_{bonds (i,j)} P[j,:] ),   S_b = sum_{j < nat} P[j,:]
// a gather of deg + 1 rows plus one rank-one term per molecule (exact, filler included).
//
// Why a third form.  The matrix-core kernels (agg.hip) multiply the full nat x nat block: at 256 atoms
// 99 % of their MFMAs multiply by 1e-9, and over small molecules with wide layers (HIV: 1250 columns per view) they re-read every
// operand row once per 16-row tile and stream at 1-1.7 TB/s.  The bond-list kernel of round 4 (sagg.hip, removed in round 6) did 64x fewer multiply-adds
// and lost anyway: a wavefront owned a molecule and gathered its neighbour rows from L2, every batch of rows one memory round trip
// behind the previous one.  Here the round trips are taken ONCE per workgroup, for everything, and the gathers are LDS reads:
//   * a workgroup owns a ROW BLOCK -- whole molecules, greedily packed to <= 256 packed rows / 16 molecules (eagcn_batch.blk, built
//     with the batch index) -- and a 32-column chunk of one view;
//   * it loads the block's rows of the operand (one 16-byte load per lane and row: 128 contiguous bytes per row), the rows' list
//     headers and the block's list entries (contiguous: the entries of molecule b live in [edge0[b], edge0[b+1])) in ONE batch of
//     independent loads, and puts them into LDS (32 KB of operand; list entries as {atom, sigma});
//   * one thread per row turns the row's list into a RECORD -- four bond weights with the row scale and the filler folded in, the
//     source rows, the self / rank-one weights (struct comment below) -- so that the row loop has no list walk and no branch;
//   * S_b: 32 groups of 8 lanes sum contiguous row ranges (one batch of LDS reads) and add them to the molecule's LDS slot;
//   * then 8 lanes own a row: 2-3 record reads, five `ds_read_b128` gathers, 24 FMAs, one 16-byte store.  BatchNorm partial sums
//     (fp64) are carried per lane and leave as one slab per row block (layer.hip bn_finalize counts the live blocks from meta[NBLK]).
// Every operand element is read from memory once and every result written once; the matrix cores are not involved (there is no
// dense block to multiply).
// Backward (one kernel): the transposed aggregation dP[j,:] = sum_{bonds (i,j)} s_i sigma_ij dY'[i,:] + s_j r dY'[j,:] + 1e-9 (G_b -
// sum_{bonds} s_i dY'[i,:]), G_b = sum_i s_i dY'[i,:], s_i = m_i / rowsum_i -- and, from the SAME gathers, the edge gradients of
// agg.hip's edge_grad_kernel: with dY' of the block in LDS, row j's own P and Y' rows in registers,
//     dA^[i,j] = <dY'[i,:], P[j,:]> ,  rowdot_i = <dY'[i,:], Y'[i,:]> ,  dU[i,j] = s_i (dA^[i,j] - rowdot_i)
// are partial sums over the workgroup's 32 columns (everything is linear in them): d w_k[c] += dU s (1 - s) by bond type through an
// LDS histogram, d self_r from the diagonal, flushed with fp64 atomics into the shared accumulator slabs (kernels.h EDGE_COPIES).
// The BatchNorm backward's second pass (layer.hip bn_bwd_apply_kernel: dY' from dH, Y' and five per-column constants) is applied to the
// rows while they are staged (AggArgs.bn_tab; Concate layers): that launch, and a write and a read of T x Fp floats, are gone -- which
// is what makes this kernel the faster BACKWARD for every Concate configuration, small molecules included (policy: lagg_use below).
// Accuracy: the filler enters as one exact rank-one term per molecule instead of nat - deg products of 1e-9 summed one by one on the
// fp32 matrix core: against float64 d att.weight is off by 1e-6 of its scale at 256-atom molecules where agg.hip is off by up to 6e-4.
#include <stdlib.h>

#include <algorithm>

#include "common.h"
#include "kernels.h"

namespace eagcn {

constexpr int LG_CW = 32;                    // columns per workgroup
constexpr int LG_LPR = LG_CW / 4;            // lanes per row (16 bytes each)
constexpr int LG_G = 256 / LG_LPR;           // row groups per workgroup (32)
constexpr int LG_U = LAGG_RB / LG_G;         // staging loads per lane (8)
constexpr int LG_ECAP = 768;                 // list entries of a block staged in LDS (the rest is read from memory)
constexpr int LG_EPT = LG_ECAP / 256;
constexpr bool LG_MERGE = true;              // transposed: edge gradients inside the row loop (false: the round-5 loop of their own)
constexpr int LG_PJ_EARLY = 4;             // transposed: P rows of the edge gradients requested in front of barrier B3 (the rest behind it: registers)

__device__ __forceinline__ void lg_fma(float4& acc, float w, const float4& v) {
    acc.x = fmaf(w, v.x, acc.x); acc.y = fmaf(w, v.y, acc.y); acc.z = fmaf(w, v.z, acc.z); acc.w = fmaf(w, v.w, acc.w);
}
__device__ __forceinline__ void lg_add(float4& acc, const float4& v) { acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w; }
__device__ __forceinline__ float lg_dot(const float4& a, const float4& b) { return fmaf(a.x, b.x, fmaf(a.y, b.y, fmaf(a.z, b.z, a.w * b.w))); }
// sum over the LG_LPR lanes of a row group (the groups are aligned 8-lane runs of a wavefront)
// (DPP operands, no LDS traffic: quad_perm [1,0,3,2], quad_perm [2,3,0,1], then row_half_mirror -- lane i of an aligned run of eight
//  meets lane 7 - i, which is in the OTHER quad and holds that quad's sum.  Same additions in the same order as three xor shuffles;
//  those compile to ds_bpermute_b32: ~150 LDS-pipe operations per lane and chunk in the transposed kernel, three dependent ones per sum)
template <int CTRL>
__device__ __forceinline__ float lg_dpp(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float lg_gsum(float v) {
    v += lg_dpp<0xB1>(v); v += lg_dpp<0x4E>(v); v += lg_dpp<0x141>(v);
    return v;
}

// list entries of a block in LDS: {atom inside its molecule, sigma, bond-type code}
struct LgLists {
    unsigned short nb[LG_ECAP];
    float w[LG_ECAP];
    unsigned char cd[LG_ECAP];
};
// ... and what overwrites them once every row's RECORD is built (transposed form; the forward has room for both):
//   forward     [0] = w_0..3 = sc (sigma_e - 1e-9)          [1] = { sc r m_i, sc, src_0..3 (bytes), meta }
//   transposed  [0] = w_0..3 = s_src (sigma_e - 1e-9)       [1] = h_0..3 = s_src sigma_e (1 - sigma_e)      [2] = { s_j, src_0..3, code_0..3, meta }
//   meta = first list entry (16 bits, relative to the block) | bonds << 16 (8 bits) | molecule << 24 (4 bits) | SLOW << 28
//   SLOW rows (more than four bonds, or -- transposed -- a self bond) take the general loop over the lists instead
constexpr uint32_t LG_SLOW = 1u << 28;
//   rows with five to eight bonds: bonds 4..7 in an OVERFLOW record of the same layout (LG_NOVF slots per block, handed out by an LDS
//   counter; meta's low 16 bits then hold the slot): the row stays in the branch-light first pass.  Beyond eight bonds, a self bond
//   (transposed) or no slot left: SLOW.
constexpr uint32_t LG_OVF = 1u << 29;
constexpr int LG_NOVF = 32;

// MULTI: the workgroup takes several column chunks per block (a.cpw > 1; the single-chunk instantiation is the round-5 kernel)
// WFUSE (transposed, Weighted_sum layers): a.src is the layer's upstream gradient; dH is formed in the staging (AggArgs.w_aw)
template <bool TRANS, bool MULTI, bool WFUSE = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 8))) void lagg_kernel(AggArgs a, EdgeArgs ed) {
    static_assert(TRANS || !WFUSE, "the upstream-gradient form belongs to the transposed kernel");
    constexpr int NREC = TRANS ? 3 : 2;
    __shared__ float4 buf[LAGG_RB][LG_LPR];          // the block's operand rows x this chunk's columns (32 KB)
    __shared__ float4 s_rec[LAGG_RB][NREC];          // row records (8 / 12 KB)
    __shared__ float4 s_ovf[LG_NOVF][NREC];          // overflow records (bonds 4..7 of the rows that have them)
    __shared__ int s_novf;
    __shared__ __attribute__((aligned(16))) unsigned char s_lists_raw[TRANS ? 16 : sizeof(LgLists)];   // forward: the staged lists
    __shared__ float s_rs[TRANS ? LAGG_RB : 1];      // transposed: s_i = m_i / rowsum_i
    __shared__ float s_rd[TRANS ? LAGG_RB : 1];      // transposed: this chunk's part of rowdot_i = <dY'_i, Y'_i>
    __shared__ unsigned char s_rm[LAGG_RB];          // molecule of the row (index inside the block)
    __shared__ float4 s_S[LAGG_MAXM][LG_LPR];        // S_b / G_b per molecule
    __shared__ float sig_s[256];
    __shared__ double st_s[TRANS ? 1 : 4][TRANS ? 1 : LG_LPR][8];   // forward: per wave: BatchNorm partial sums of a lane's four columns
    __shared__ float4 s_bn[TRANS ? (WFUSE ? 4 : 3) : 1][LG_LPR];   // transposed + BatchNorm fusion: three (Weighted_sum: four) constants per column of this chunk
    __shared__ double h_s[TRANS ? 264 : 1];          // transposed: bond-type histogram of d w_k, [256] = d self_r
    // transposed: the lists live in the record array until the records are built (LDS: 51 KB = three workgroups per CU either way)
    static_assert(sizeof(LgLists) <= sizeof(float4) * LAGG_RB * 3, "lists alias the transposed record array");
    LgLists& L = *reinterpret_cast<LgLists*>(TRANS ? reinterpret_cast<unsigned char*>(&s_rec[0][0]) : s_lists_raw);
    const eagcn_batch& bt = a.bt;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // a workgroup takes a.cpw CONSECUTIVE 32-column chunks of its view for every block it owns (round 6): the block's lists and row
    // records do not depend on the columns, so they are staged and built ONCE per block and the chunks only repeat the operand
    // staging, the column sums and the row passes (with one chunk per workgroup a 1250-column view rebuilt the same records forty times)
    const int k = blockIdx.x / a.nchunk, grp = blockIdx.x - k * a.nchunk;         // (a.nchunk: chunk GROUPS per view)
    const int wk = a.vc.off[k + 1] - a.vc.off[k];                    // padded width of the view (a multiple of 16)
    const int cc_lo = MULTI ? grp * a.cpw : grp;
    if (cc_lo * LG_CW >= wk) return;                                  // (uniform)
    const int cc_hi = MULTI ? min(cc_lo + a.cpw, (wk + LG_CW - 1) / LG_CW) : cc_lo + 1;
    // the first block's record is requested together with the block count it is checked against (the index is inside the record
    // array's capacity -- one record per molecule and gridDim.y <= B): one memory round trip at the head of the workgroup, not two
    const int4* blk4 = reinterpret_cast<const int4*>(bt.blk);
    int4 b0n = blk4[2 * blockIdx.y], b1n = blk4[2 * blockIdx.y + 1];
    const int nblk = bt.meta[EAGCN_META_NBLK];
    const int nlog = dev_n(bt);
    const float r = a.rsig[k];
    sig_s[tid] = a.sig[k * 256 + tid];                               // (made visible by the first barrier of the block loop)
    const int l = tid & (LG_LPR - 1);
    int g = tid / LG_LPR;                                             // (row group; re-declared opaque per block below)
    int col, c0, c0s;                                                 // this lane's first column inside the view / the matrix, per chunk
    bool col_ok;
    auto set_chunk = [&](int cc) __attribute__((always_inline)) {
        col = cc * LG_CW + 4 * l;
        col_ok = col < wk;
        c0 = a.vc.off[k] + col;
        c0s = col_ok ? c0 : a.vc.off[k];                              // (a legal column for the lanes beyond the view's width)
    };
    const float* rsk = a.rscale + (size_t)k * bt.T;
    const int2* ptrs = reinterpret_cast<const int2*>(TRANS ? bt.col_ptr : bt.row_ptr);
    const int32_t* nbr = TRANS ? bt.tnbr : bt.nbr;
    const uint64_t* codes = TRANS ? bt.tcode : bt.ecode;
    if constexpr (TRANS) { h_s[tid] = 0.0; if (tid < 8) h_s[256 + tid] = 0.0; }
    double dr_acc = 0.0;
    const int dbg = a.xcd;                                            // (probe mask, EAGCN_LAGG_DBG: wrong results)
    const int4* rinfo = reinterpret_cast<const int4*>(bt.row_info);
    // transposed with the BatchNorm backward's second pass folded in (AggArgs.bn_tab): this lane's five per-column constants
    const bool fuse_bn = TRANS && a.bn_tab != nullptr;
    // dY' = sc (dH - c1 - (Y' - mu) inv c2) (bn_bwd_apply_kernel) as A dH + Bc Y' + Cc: three constants per column of the chunk, kept in
    // LDS (twenty registers across the block loop otherwise); the regrouping moves the result by an ulp of its largest term.  With
    // several chunks per workgroup the NEXT chunk's constants are put there behind a chunk's last barrier (thirty-two lanes, one
    // round trip to L2 beside the edge loop; holding them in registers from the head of the chunk spilled).
    auto bn_consts = [&](int cc) __attribute__((always_inline)) {
        const int cl = cc * LG_CW + tid < wk ? a.vc.off[k] + cc * LG_CW + tid : a.vc.off[k];
        const float sc = a.bn_tab[(size_t)BN_SC * a.bn_fp + cl], mu = a.bn_tab[(size_t)BN_MU * a.bn_fp + cl];
        const float iv = a.bn_tab[(size_t)BN_INV * a.bn_fp + cl], c1 = a.bn_cc[cl], c2 = a.bn_cc[a.bn_fp + cl];
        float* sb = reinterpret_cast<float*>(&s_bn[0][0]);
        sb[tid] = sc;
        sb[LG_CW + tid] = -sc * iv * c2;
        sb[2 * LG_CW + tid] = sc * (mu * iv * c2 - c1);
        if constexpr (WFUSE) sb[3 * LG_CW + tid] = a.bn_tab[(size_t)BN_SH * a.bn_fp + cl];
    };
    // (Ave_multi_view.weight is one scalar per view, layers.py:423: uniform for the workgroup.  A fifth constant row in LDS would be
    //  the 256 bytes that cost the third workgroup per CU: 3 x 54 472 bytes fill the CU's 160 KB to 424 bytes)
    const float w_ave = WFUSE ? a.w_aw[a.vc.off[k]] : 0.0f;
    const uint64_t wseed = WFUSE && a.w_drop ? (a.w_seed_dev ? *a.w_seed_dev : a.w_seed) : 0ull;
    if constexpr (TRANS) {
        if (fuse_bn && tid < LG_CW) bn_consts(cc_lo);
    }
    // A workgroup takes the blocks q, q + gridDim.y, ... (the grid's y extent is an estimate of the block count).  Measured and
    // dropped: a software pipeline over a workgroup's blocks (the next block's rows in flight into registers while this one is worked
    // on, a persistent grid of three workgroups per CU): 212 / 238 registers = two workgroups per CU instead of three, and slower.
    for (int q = blockIdx.y; q < nblk; q += gridDim.y) {
        // the block: {first molecule, molecules, first packed row, rows} {first list entry, entries} -- one dependent load, then everything
        const int4 b0 = b0n, b1 = b1n;
        asm volatile("" : "+v"(g));                                   // (a workgroup has ONE block as a rule: per-thread row indices and LDS
                                                                      //  addresses hoisted out of this loop only cost registers -- and spilled)
        if (q + (int)gridDim.y < nblk) { b0n = blk4[2 * (q + gridDim.y)]; b1n = blk4[2 * (q + gridDim.y) + 1]; }      // (only when the grid was an underestimate)
        const int m0 = b0.x, R0 = b0.z, rows = min(b0.w, LAGG_RB), E0 = b1.x, ne = b1.y;
        if (rows <= 0) {                                              // (uniform) nothing stored: the slab still has to be defined
            if constexpr (!TRANS) {
                const int fp = a.vc.off[a.vc.K];
                for (int cc = cc_lo; cc < cc_hi; ++cc)
                    if (tid < LG_CW && cc * LG_CW + tid < wk)
                        *reinterpret_cast<double2*>(a.stats + ((size_t)q * fp + a.vc.off[k] + cc * LG_CW + tid) * 2) = make_double2(0.0, 0.0);
            }
            continue;
        }
        const int nst = min(ne, LG_ECAP);
        // ---- what does not depend on the columns: row descriptors, list headers, list entries -> LDS -> one RECORD per row.  One chunk
        //      per workgroup: requested in the same batch as the operand rows and built beside their staging (one chain of round trips).
        //      Several chunks (MULTI): a phase of its own in front of the chunk loop -- one more round trip per BLOCK, and nothing of it
        //      is live inside the chunk loop (interleaved with the first chunk it cost the transposed kernel 44 bytes of scratch).
        int2 pt;
        float rsv;
        int4 ri;
        int e_jn[LG_EPT];
        uint64_t e_cd[LG_EPT];
        float4 rec[NREC];
        int my_mol, my_off;
        auto hdr_loads = [&]() __attribute__((always_inline)) {
            const int tr = R0 + min(tid, rows - 1);
            pt = ptrs[tr];
            rsv = TRANS ? rsk[tr] : bt.row_m[tr];                     // forward: m_i; transposed: s_j
            ri = rinfo[tr];                                           // {molecule, atom, nat, first row of the molecule}
        };
        auto list_loads = [&]() __attribute__((always_inline)) {
#pragma unroll
            for (int u = 0; u < LG_EPT; ++u) {
                const int ec = E0 + min(tid + 256 * u, nst - 1);
                e_jn[u] = nbr[ec];
                e_cd[u] = codes[ec];
            }
        };
        auto lists_to_lds = [&]() __attribute__((always_inline)) {
            my_mol = min(max(ri.x - m0, 0), LAGG_MAXM - 1);           // (of row `tid`, tid < rows)
            my_off = ri.w - R0;
            if (tid == 0) s_novf = 0;
            if (tid < rows) {
                s_rm[tid] = (unsigned char)my_mol;
                if constexpr (TRANS) s_rs[tid] = rsv;
            }
            if (nst > 0) {
#pragma unroll
                for (int u = 0; u < LG_EPT; ++u) {
                    const int e = tid + 256 * u;
                    if (e < nst) {
                        const uint32_t c = (uint32_t)(e_cd[u] >> (8 * k)) & 255u;
                        L.nb[e] = (unsigned short)e_jn[u];
                        L.w[e] = sig_s[c];
                        L.cd[e] = (unsigned char)c;
                    }
                }
            }
        };
        // entry `el` of the block's lists: {atom inside its molecule, sigma, code}; from LDS while the lists are there, else from memory
        auto entry = [&](int el, bool lds_ok, int& jn, float& w, uint32_t& c) __attribute__((always_inline)) {
            if (lds_ok && el < LG_ECAP) {
                jn = L.nb[el]; w = L.w[el]; c = L.cd[el];
            } else {
                jn = nbr[E0 + el];
                c = (uint32_t)(codes[E0 + el] >> (8 * k)) & 255u;
                w = sig_s[c];
            }
        };
        auto build_record = [&]() __attribute__((always_inline)) {
            // ---- the row RECORDS: thread t builds row t's (header comment of the struct above).  The first version of this kernel read a
            //      row's state from five LDS arrays and walked its list entries in a dynamic loop with a running rowsum and a second
            //      accumulator for the filler -- it was bound by instruction ISSUE (SQ_ACTIVE 30 % per wave at three waves per SIMD,
            //      profiles/r05_lagg_sq.txt).  With the scale and the filler folded into the weights,
            //          forward      y_i  = sum_e w_e P[src_e] + (sc r m_i) P[i] + (sc 1e-9) S_b
            //          transposed   dP_j = sum_e w_e Z[src_e] + (r s_j) Z[j] + 1e-9 G_b ,   d w[code_e] += h_e (<Z[src_e], P_j> - rowdot_src)
            //      the row loop is two or three LDS reads, five gathers and a few dozen FMAs without a branch.
            const int first = pt.x - E0, cnt = (dbg & 2) ? 0 : pt.y;
            if (tid < rows) {
                float we[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, he[TRANS ? 8 : 1] = {0.f};
                uint32_t srcs = 0u, cds = 0u, srcs2 = 0u, cds2 = 0u, slow = cnt > 8 ? LG_SLOW : 0u;
                float wsum = 0.0f;
                if constexpr (TRANS) {
    #pragma unroll
                    for (int e = 1; e < 8; ++e) he[e] = 0.0f;
                }
                auto take = [&](int e, int jn, float w, uint32_t c, float ss) __attribute__((always_inline)) {
                    const int src = min(my_off + jn, LAGG_RB - 1);
                    if constexpr (TRANS) { if (src == tid) slow = LG_SLOW; }      // (a self bond: the diagonal of the edge gradients is this entry)
                    const float wv = ss * (w - TINY), hv = TRANS ? ss * w * (1.0f - w) : 0.0f;
    #pragma unroll
                    for (int q = 0; q < 8; ++q)                           // (constant register indices)
                        if (q == e) { we[q] = wv; if constexpr (TRANS) he[q] = hv; }
                    if (e < 4) { srcs |= (uint32_t)src << (8 * e); if constexpr (TRANS) cds |= c << (8 * e); }
                    else { srcs2 |= (uint32_t)src << (8 * (e - 4)); if constexpr (TRANS) cds2 |= c << (8 * (e - 4)); }
                };
                if (first + 8 <= LG_ECAP) {
                    // the row's first eight list slots in ONE batch of LDS reads (slots beyond its count: any legal slot, not used), the scales
                    // of their source rows in a second: two LDS round trips per row instead of two per bond
                    int jn8[8]; float w8[8], ss8[8]; uint32_t c8[8];
    #pragma unroll
                    for (int e = 0; e < 8; ++e) { jn8[e] = L.nb[first + e]; w8[e] = L.w[first + e]; c8[e] = L.cd[first + e]; }
    #pragma unroll
                    for (int e = 0; e < 8; ++e) ss8[e] = TRANS ? s_rs[min(my_off + jn8[e], LAGG_RB - 1)] : 1.0f;
    #pragma unroll
                    for (int e = 0; e < 8; ++e)
                        if (e < cnt) { wsum += w8[e]; take(e, jn8[e], w8[e], c8[e], ss8[e]); }
                    for (int e = 8; e < cnt; ++e) {                       // (a SLOW row: only its row sum is needed here)
                        int jn; float w; uint32_t c;
                        entry(first + e, true, jn, w, c);
                        wsum += w;
                    }
                } else {
                    for (int e = 0; e < cnt; ++e) {
                        int jn; float w; uint32_t c;
                        entry(first + e, true, jn, w, c);
                        wsum += w;
                        if (e < 8) take(e, jn, w, c, TRANS ? s_rs[min(my_off + jn, LAGG_RB - 1)] : 1.0f);
                        else if constexpr (TRANS) { if (min(my_off + jn, LAGG_RB - 1) == tid) slow = LG_SLOW; }
                    }
                }
                for (int e = min(cnt, 4); e < 4; ++e) srcs |= (uint32_t)tid << (8 * e);       // (weight 0: any legal row)
                for (int e = min(max(cnt, 4), 8); e < 8; ++e) srcs2 |= (uint32_t)tid << (8 * (e - 4));
                int slot = 0;
                if (cnt > 4 && !slow) {
                    slot = atomicAdd(&s_novf, 1);
                    if (slot >= LG_NOVF) slow = LG_SLOW;
                }
                const bool ovf = cnt > 4 && !slow;
                const uint32_t meta = (uint32_t)((ovf ? slot : first) & 0xFFFF) | ((uint32_t)min(cnt, 255) << 16) | ((uint32_t)my_mol << 24) | slow | (ovf ? LG_OVF : 0u);
                if constexpr (!TRANS) {
                    const float d = wsum + r * rsv + TINY * (float)(nlog - cnt);           // rowsum: sum sigma + r m_i + 1e-9 (columns without a bond)
                    const float sc = rsv > 0.0f ? 1.0f / d : 0.0f;
                    if (cc_lo == 0) a.rscale[(size_t)k * bt.T + R0 + tid] = sc;
                    rec[0] = make_float4(sc * we[0], sc * we[1], sc * we[2], sc * we[3]);
                    rec[1] = make_float4(sc * r * rsv, sc, __uint_as_float(srcs), __uint_as_float(meta));
                    if (ovf) {
                        s_ovf[slot][0] = make_float4(sc * we[4], sc * we[5], sc * we[6], sc * we[7]);
                        s_ovf[slot][1] = make_float4(0.f, 0.f, __uint_as_float(srcs2), 0.f);
                    }
                } else {
                    rec[0] = make_float4(we[0], we[1], we[2], we[3]);
                    rec[1] = make_float4(he[0], he[1], he[2], he[3]);
                    rec[NREC - 1] = make_float4(rsv, __uint_as_float(srcs), __uint_as_float(cds), __uint_as_float(meta));
                    if (ovf) {
                        s_ovf[slot][0] = make_float4(we[4], we[5], we[6], we[7]);
                        s_ovf[slot][1] = make_float4(he[4], he[5], he[6], he[7]);
                        s_ovf[slot][NREC - 1] = make_float4(0.f, __uint_as_float(srcs2), __uint_as_float(cds2), 0.f);
                    }
                }
            }
        };
        auto write_record = [&]() __attribute__((always_inline)) {
            if (tid < rows) {
#pragma unroll
                for (int i = 0; i < NREC; ++i) s_rec[tid][i] = rec[i];
            }
        };
        if constexpr (MULTI) {
            hdr_loads();
            if (nst > 0) list_loads();                                // (uniform)
            __syncthreads();                                          // R1: the LDS of the previous block is free
            lists_to_lds();
            __syncthreads();                                          // R2: lists and scales are in LDS
            build_record();
            if constexpr (TRANS) __syncthreads();                     // (the records take the lists' place)
            write_record();                                           // (made visible by the chunk loop's barriers)
        }
        for (int cc = cc_lo; cc < cc_hi; ++cc) {                      // ---- chunks of the group
        set_chunk(cc);
        asm volatile("" : "+v"(g));                                   // (row indices / LDS addresses are NOT hoisted out of the chunk loop either)
        const bool bn_next = MULTI && TRANS && fuse_bn && tid < LG_CW;

        // ---- ONE batch of independent loads: operand rows, (transposed) the rows' own Y', row descriptors, list headers, list entries.
        //      Every load is unconditional on a clamped address (a load under a per-lane condition compiles to a branch and, behind it,
        //      a wait per load); what a lane must not use is zeroed afterwards.
        float4 v[LG_U], yv[TRANS ? LG_U : 1];
#pragma unroll
        for (int u = 0; u < LG_U; ++u) {
            const int rc = min(g + LG_G * u, rows - 1);
            v[u] = *reinterpret_cast<const float4*>(a.src + (size_t)(R0 + rc) * a.lds + (WFUSE ? (col_ok ? col : 0) : c0s));
            if constexpr (TRANS) yv[u] = *reinterpret_cast<const float4*>(ed.Y + (size_t)(R0 + rc) * ed.ld + c0s);
        }
        if constexpr (!MULTI) {
            hdr_loads();
            if (nst > 0) list_loads();                                // (uniform)
        }
        // Weighted_sum form: the dropout draws of this lane's 8 x 4 elements as ONE mask register, hashed while the loads are in flight
        uint32_t keep = 0xFFFFFFFFu;
        if constexpr (WFUSE) {
            if (a.w_drop) {                                           // (uniform)
                keep = 0u;
                const uint32_t t16 = a.w_thr >> 16;
#pragma unroll
                for (int u = 0; u < LG_U; ++u) {
                    // (drop_scale4 of element (row, c0s); the row is NOT clamped -- rows beyond the block's are not stored, and eight clamped
                    //  row indices held for this cost the kernel eight registers and its third workgroup per CU)
                    const uint64_t z = rng_u64(wseed, ((uint64_t)(R0 + g + LG_G * u) * a.bn_fp + c0s) >> 2);
                    const uint32_t lo = (uint32_t)z, hi = (uint32_t)(z >> 32);
                    keep |= ((lo & 0xFFFFu) >= t16 ? 1u : 0u) << (4 * u);
                    keep |= ((lo >> 16) >= t16 ? 2u : 0u) << (4 * u);
                    keep |= ((hi & 0xFFFFu) >= t16 ? 4u : 0u) << (4 * u);
                    keep |= ((hi >> 16) >= t16 ? 8u : 0u) << (4 * u);
                }
            }
        }
        __syncthreads();                                              // B1: the LDS of the previous block / chunk is free
        if (tid < LAGG_MAXM * LG_LPR) {
            float z;
            asm volatile("v_mov_b32 %0, 0" : "=v"(z));                // (made here: hoisted out of the block loop the zero vector is spilled)
            (&s_S[0][0])[tid] = make_float4(z, z, z, z);
        }
#pragma unroll
        for (int u = 0; u < LG_U; ++u) {
            const int rr = g + LG_G * u;                              // (consecutive groups = consecutive rows: no LDS bank conflicts)
            const bool mine = rr < rows;
            if constexpr (TRANS) {
                if (fuse_bn) {                                        // dY' from dH and Y'
                    if constexpr (WFUSE) {
                        // dH of this view from the upstream gradient, exactly as the reduction pass formed it (layer.hip bn_bwd_reduce_kernel):
                        // dH = relu'(sc Y' + sh) keep (up ave_w) / (1 - p).  Three steps, each with its own constants read from LDS and an
                        // order fixed by empty asm statements -- with all five constant vectors of a row in flight at once (twenty
                        // registers) the kernel lost its third workgroup per CU; one select per element and no branch (`h > 0 ? x : 0`
                        // per component put the LDS reads under exec-mask branches)
                        int lq = l;
                        asm volatile("" : "+v"(lq));
                        uint32_t mb;
                        {
                            const float4 bA = s_bn[0][lq], sh = s_bn[3][lq];
                            const uint32_t kb = keep >> (4 * u);
                            mb = ((yv[u].x * bA.x + sh.x > 0.0f) ? (kb & 1u) : 0u) | ((yv[u].y * bA.y + sh.y > 0.0f) ? (kb & 2u) : 0u) |
                                 ((yv[u].z * bA.z + sh.z > 0.0f) ? (kb & 4u) : 0u) | ((yv[u].w * bA.w + sh.w > 0.0f) ? (kb & 8u) : 0u);
                        }
                        asm volatile("" : "+v"(mb));
                        {
                            const float ik = a.w_drop ? a.w_inv_keep : 1.0f;
                            v[u].x = (v[u].x * w_ave) * ((mb & 1u) ? ik : 0.0f);
                            v[u].y = (v[u].y * w_ave) * ((mb & 2u) ? ik : 0.0f);
                            v[u].z = (v[u].z * w_ave) * ((mb & 4u) ? ik : 0.0f);
                            v[u].w = (v[u].w * w_ave) * ((mb & 8u) ? ik : 0.0f);
                        }
                        asm volatile("" : "+v"(v[u].x), "+v"(v[u].y), "+v"(v[u].z), "+v"(v[u].w), "+v"(lq));
                        const float4 bA = s_bn[0][lq], bB = s_bn[1][lq], bC = s_bn[2][lq];
                        v[u].x = fmaf(bA.x, v[u].x, fmaf(bB.x, yv[u].x, bC.x));
                        v[u].y = fmaf(bA.y, v[u].y, fmaf(bB.y, yv[u].y, bC.y));
                        v[u].z = fmaf(bA.z, v[u].z, fmaf(bB.z, yv[u].z, bC.z));
                        v[u].w = fmaf(bA.w, v[u].w, fmaf(bB.w, yv[u].w, bC.w));
                    } else {
                        const float4 bA = s_bn[0][l], bB = s_bn[1][l], bC = s_bn[2][l];
                        v[u].x = fmaf(bA.x, v[u].x, fmaf(bB.x, yv[u].x, bC.x));
                        v[u].y = fmaf(bA.y, v[u].y, fmaf(bB.y, yv[u].y, bC.y));
                        v[u].z = fmaf(bA.z, v[u].z, fmaf(bB.z, yv[u].z, bC.z));
                        v[u].w = fmaf(bA.w, v[u].w, fmaf(bB.w, yv[u].w, bC.w));
                    }
                }
            }
            if (!col_ok) { v[u] = make_float4(0.f, 0.f, 0.f, 0.f); if constexpr (TRANS) yv[u] = make_float4(0.f, 0.f, 0.f, 0.f); }
            if (mine) buf[rr][l] = v[u];
            if constexpr (TRANS) {
                const float d = lg_gsum(lg_dot(v[u], yv[u]));         // this chunk's part of rowdot_i (the operands are in registers)
                if (mine && l == 0) s_rd[rr] = d;
            }
        }
        if constexpr (!MULTI) lists_to_lds();
        __syncthreads();                                              // B2: rows, lists, scales are in LDS
        // ---- S_b (forward) / G_b = sum_i s_i dY'_i (transposed) per molecule: group g sums the contiguous rows [g per, (g + 1) per) -- all
        //      of them read from LDS in ONE batch, then added up in registers (row by row behind the data-dependent molecule test the
        //      reads cost 4 us of an 18 us launch at configs[1]; contiguous OWNERSHIP of rows -- sums straight from the staging
        //      registers -- puts the eight groups of a wave on the same banks in every other phase: C5 11.0 -> 12.2 ms)
        if (!(dbg & 1)) {
            const int per = (rows + LG_G - 1) / LG_G, ra = g * per;
            int mu[LG_U];
            float wu[TRANS ? LG_U : 1];
            float4 bu[LG_U];
#pragma unroll
            for (int u = 0; u < LG_U; ++u) {
                const int rc = min(ra + u, rows - 1);
                mu[u] = s_rm[rc];
                bu[u] = buf[rc][l];
                if constexpr (TRANS) wu[u] = s_rs[rc];
            }
            float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
            int cur = -1;
            auto flush = [&]() {
                if (cur >= 0) {
                    float* dst = reinterpret_cast<float*>(&s_S[cur][l]);
                    atomicAdd(dst + 0, acc.x); atomicAdd(dst + 1, acc.y); atomicAdd(dst + 2, acc.z); atomicAdd(dst + 3, acc.w);
                }
            };
#pragma unroll
            for (int u = 0; u < LG_U; ++u) {
                if (u < per && ra + u < rows) {
                    if (mu[u] != cur) { flush(); acc = make_float4(0.f, 0.f, 0.f, 0.f); cur = mu[u]; }
                    if constexpr (TRANS) lg_fma(acc, wu[u], bu[u]); else lg_add(acc, bu[u]);
                }
            }
            flush();
        }
        if constexpr (!MULTI) {
            build_record();
            if constexpr (TRANS) __syncthreads();                     // B2b: every thread is done with the lists: the records take their place
            write_record();
        }
        // transposed: row j's own P row for the edge gradients: all eight of a group requested HERE, in front of the barrier, and used by
        // a loop of their own below that issues no store (with loads and stores in one loop the compiler cannot count the memory
        // operations in flight and waits for ALL of them in every iteration: 3 900 cycles per row, measured with s_memtime stamps)
        auto pj_load = [&](int u) __attribute__((always_inline)) {
            int rc = R0 + min(g + LG_G * u, rows - 1);
            asm volatile("" : "+v"(rc));                              // (else the eight 64-bit row offsets of the staging loads stay live for this)
            return *reinterpret_cast<const float4*>(ed.P + (size_t)rc * ed.ld + c0s);
        };
        float4 pjv[TRANS ? LG_U : 1];
        if constexpr (TRANS) {                                        // (the first half here, the second at the head of the edge loop:
#pragma unroll                                                        //  all eight in front of the barrier are five registers too many)
            for (int u = 0; u < LG_PJ_EARLY; ++u) pjv[u] = pj_load(u);
        }
        __syncthreads();                                              // B3: records and S_b / G_b are complete
        if constexpr (TRANS) { if (bn_next) bn_consts(cc + 1 < cc_hi ? cc + 1 : cc_lo); }     // (the staging of THIS chunk has read s_bn)
        if constexpr (TRANS && LG_MERGE) {
#pragma unroll
            for (int u = LG_PJ_EARLY; u < LG_U; ++u) pjv[u] = pj_load(u);
        }
        if constexpr (TRANS && !LG_MERGE) {
            // ---- edge gradients of the rows with a record (this chunk's columns): d w[code_e] += h_e (<Z[src_e], P_j> - rowdot_src), the
            //      diagonal into d self_r.  Lane e (< 4) of the row's eight adds bond e's term, lane 4 the diagonal's.
            const int nu = (rows + LG_G - 1) / LG_G;
#pragma unroll
            for (int u = LG_PJ_EARLY; u < LG_U; ++u) pjv[u] = pj_load(u);
#pragma unroll
            for (int u = 0; u < LG_U; ++u) {
                if (u >= nu) break;                                   // (uniform; rows beyond the block's / of the second pass: nothing added)
                const int rr = min(g + LG_G * u, rows - 1);
                const float4 r1 = s_rec[rr][1], rl = s_rec[rr][2];
                const uint32_t srcs = __float_as_uint(rl.y), cds = __float_as_uint(rl.z), meta = __float_as_uint(rl.w);
                const bool fast = g + LG_G * u < rows && !(meta & LG_SLOW);
                const float4 pj = col_ok ? pjv[u] : make_float4(0.f, 0.f, 0.f, 0.f);
                const float4 self = buf[rr][l];
                const float4 v0 = buf[srcs & 255u][l], v1 = buf[(srcs >> 8) & 255u][l], v2 = buf[(srcs >> 16) & 255u][l], v3 = buf[srcs >> 24][l];
                const float g0 = lg_gsum(lg_dot(v0, pj)), g1 = lg_gsum(lg_dot(v1, pj)), g2 = lg_gsum(lg_dot(v2, pj)), g3 = lg_gsum(lg_dot(v3, pj));
                const float gs = lg_gsum(lg_dot(self, pj));
                const int e4 = l & 3;
                const float hv = e4 == 0 ? r1.x : e4 == 1 ? r1.y : e4 == 2 ? r1.z : r1.w;
                const float gv = e4 == 0 ? g0 : e4 == 1 ? g1 : e4 == 2 ? g2 : g3;
                const uint32_t cv = (cds >> (8 * e4)) & 255u, sv = (srcs >> (8 * e4)) & 255u;
                if (fast && l < 4 && hv != 0.0f && cv) atomicAdd(&h_s[cv], (double)hv * ((double)gv - (double)s_rd[sv]));
                if (fast && l == 4) dr_acc += (double)rl.x * ((double)gs - (double)s_rd[rr]);
                if (fast && (meta & LG_OVF)) {                        // (bonds 4..7: same, from the overflow record)
                    const int slot = (int)(meta & 0xFFFFu);
                    const float4 q1 = s_ovf[slot][1], ql = s_ovf[slot][2];
                    const uint32_t sr2 = __float_as_uint(ql.y), cd2 = __float_as_uint(ql.z);
                    const float4 b0 = buf[sr2 & 255u][l], b1 = buf[(sr2 >> 8) & 255u][l], b2 = buf[(sr2 >> 16) & 255u][l], b3 = buf[sr2 >> 24][l];
                    const float f0 = lg_gsum(lg_dot(b0, pj)), f1 = lg_gsum(lg_dot(b1, pj)), f2 = lg_gsum(lg_dot(b2, pj)), f3 = lg_gsum(lg_dot(b3, pj));
                    const float hv2 = e4 == 0 ? q1.x : e4 == 1 ? q1.y : e4 == 2 ? q1.z : q1.w;
                    const float gv2 = e4 == 0 ? f0 : e4 == 1 ? f1 : e4 == 2 ? f2 : f3;
                    const uint32_t cv2 = (cd2 >> (8 * e4)) & 255u, sv2 = (sr2 >> (8 * e4)) & 255u;
                    if (l < 4 && hv2 != 0.0f && cv2) atomicAdd(&h_s[cv2], (double)hv2 * ((double)gv2 - (double)s_rd[sv2]));
                }
            }
        }
        double s1[4] = {0.0, 0.0, 0.0, 0.0}, s2[4] = {0.0, 0.0, 0.0, 0.0};
        // ---- the rows: 8 lanes own a row -----------------------------------------------------------------------------------------------
        // Two passes over the group's rows.  The FIRST takes the rows with a complete record and contains no memory load at all: LDS reads,
        // FMAs, stores.  Rows with more than four bonds (transposed: or a self bond) are left to the SECOND pass, whose general loop
        // reads list entries (and, transposed, the row's P) from memory.  In ONE loop the compiler cannot count the memory operations
        // in flight across the rare branch and waits for ALL of them -- the previous row's stores included -- in every iteration:
        // 1 400 (forward) / 2 800 (transposed) cycles per row, measured with s_memtime stamps.
        auto finish = [&](int rr, const float4& y) __attribute__((always_inline)) {
            if constexpr (!TRANS) {
                s1[0] += (double)y.x; s2[0] += (double)y.x * (double)y.x;
                s1[1] += (double)y.y; s2[1] += (double)y.y * (double)y.y;
                s1[2] += (double)y.z; s2[2] += (double)y.z * (double)y.z;
                s1[3] += (double)y.w; s2[3] += (double)y.w * (double)y.w;
                if (col_ok && !(dbg & 4)) *reinterpret_cast<float4*>(a.dst + (size_t)(R0 + rr) * a.ldd + c0) = y;
            } else {
                if (col_ok && !(dbg & 4)) {
                    int cs = c0;
                    asm volatile("" : "+v"(cs));                      // (no per-lane 64-bit store bases held across the block loop)
                    if (a.planes.p) bx_store4(a.planes, R0 + rr, cs, y);
                    else *reinterpret_cast<float4*>(a.dst + (size_t)(R0 + rr) * a.ldd + cs) = y;
                }
            }
        };
        bool any_slow = false;
        {
            // (forward: the NEXT row's record is read while this row's gathers are in flight; the transposed kernel has no registers to
            //  spare for that at three workgroups per CU)
            constexpr bool AHEAD = !TRANS;
            float4 nrec[NREC];
            if constexpr (AHEAD) {
#pragma unroll
                for (int i = 0; i < NREC; ++i) nrec[i] = s_rec[min(g, rows - 1)][i];
            }
            // (no divergent branch in this loop but the one around a row's store: the trip count is uniform, rows beyond the block's
            //  and rows of the second pass are computed on clamped indices and not stored)
            const int nu = (rows + LG_G - 1) / LG_G;
#pragma unroll
            for (int u = 0; u < LG_U; ++u) {
                if (u >= nu) break;                                   // (uniform)
                const int rr = min(g + LG_G * u, rows - 1);
                const bool valid = g + LG_G * u < rows;
                if constexpr (!AHEAD) {
#pragma unroll
                    for (int i = 0; i < NREC; ++i) nrec[i] = s_rec[rr][i];
                }
                const float4 r0 = nrec[0], r1 = nrec[1], rl = nrec[NREC - 1];
                if (AHEAD && u + 1 < LG_U) {
#pragma unroll
                    for (int i = 0; i < NREC; ++i) nrec[i] = s_rec[min(rr + LG_G, rows - 1)][i];
                }
                const uint32_t srcs = __float_as_uint(TRANS ? rl.y : rl.z), meta = __float_as_uint(rl.w);
                const bool fast = valid && !(meta & LG_SLOW);
                any_slow = any_slow || (valid && (meta & LG_SLOW));
                const float4 self = buf[rr][l];
                const float4 S = s_S[(meta >> 24) & 15u][l];
                const float4 v0 = buf[srcs & 255u][l], v1 = buf[(srcs >> 8) & 255u][l], v2 = buf[(srcs >> 16) & 255u][l], v3 = buf[srcs >> 24][l];
                const float ws = TRANS ? r * rl.x : r1.x;                      // weight of the row's own operand row
                const float wS = TRANS ? TINY : r1.y * TINY;                   // ... and of the molecule's column sum
                float4 y;
                y.x = fmaf(r0.x, v0.x, fmaf(r0.y, v1.x, fmaf(r0.z, v2.x, fmaf(r0.w, v3.x, fmaf(ws, self.x, wS * S.x)))));
                y.y = fmaf(r0.x, v0.y, fmaf(r0.y, v1.y, fmaf(r0.z, v2.y, fmaf(r0.w, v3.y, fmaf(ws, self.y, wS * S.y)))));
                y.z = fmaf(r0.x, v0.z, fmaf(r0.y, v1.z, fmaf(r0.z, v2.z, fmaf(r0.w, v3.z, fmaf(ws, self.z, wS * S.z)))));
                y.w = fmaf(r0.x, v0.w, fmaf(r0.y, v1.w, fmaf(r0.z, v2.w, fmaf(r0.w, v3.w, fmaf(ws, self.w, wS * S.w)))));
                if constexpr (TRANS && LG_MERGE) {
                    // the row's edge gradients from the SAME gathers (round 6: the separate loop read the record, the row and its four
                    // sources from LDS a second time -- 7 of 17 16-byte LDS reads per row and lane, and the LDS pipe is what
                    // workgroups that share a CU compete for): d w[code_e] += h_e (<Z[src_e], P_j> - rowdot_src), lane e (< 4) of
                    // the row's eight adds bond e's term, lane 4 the diagonal's (into d self_r)
                    const float4 pj = col_ok ? pjv[u] : make_float4(0.f, 0.f, 0.f, 0.f);
                    const uint32_t cds = __float_as_uint(rl.z);
                    const float g0 = lg_gsum(lg_dot(v0, pj)), g1 = lg_gsum(lg_dot(v1, pj)), g2 = lg_gsum(lg_dot(v2, pj)), g3 = lg_gsum(lg_dot(v3, pj));
                    const float gs = lg_gsum(lg_dot(self, pj));
                    const int e4 = l & 3;
                    const float hv = e4 == 0 ? r1.x : e4 == 1 ? r1.y : e4 == 2 ? r1.z : r1.w;
                    const float gv = e4 == 0 ? g0 : e4 == 1 ? g1 : e4 == 2 ? g2 : g3;
                    const uint32_t cv = (cds >> (8 * e4)) & 255u, sv = (srcs >> (8 * e4)) & 255u;
                    if (fast && l < 4 && hv != 0.0f && cv) atomicAdd(&h_s[cv], (double)hv * ((double)gv - (double)s_rd[sv]));
                    if (fast && l == 4) dr_acc += (double)rl.x * ((double)gs - (double)s_rd[rr]);
                }
                if (fast && (meta & LG_OVF)) {                        // (bonds 4..7)
                    const int slot = (int)(meta & 0xFFFFu);
                    const float4 q0 = s_ovf[slot][0];
                    const uint32_t sr2 = __float_as_uint(TRANS ? s_ovf[slot][NREC - 1].y : s_ovf[slot][NREC - 1].z);
                    const float4 b0 = buf[sr2 & 255u][l], b1 = buf[(sr2 >> 8) & 255u][l], b2 = buf[(sr2 >> 16) & 255u][l], b3 = buf[sr2 >> 24][l];
                    lg_fma(y, q0.x, b0); lg_fma(y, q0.y, b1); lg_fma(y, q0.z, b2); lg_fma(y, q0.w, b3);
                    if constexpr (TRANS && LG_MERGE) {
                        const float4 pj = col_ok ? pjv[u] : make_float4(0.f, 0.f, 0.f, 0.f);
                        const float4 q1 = s_ovf[slot][1];
                        const uint32_t cd2 = __float_as_uint(s_ovf[slot][2].z);
                        const float f0 = lg_gsum(lg_dot(b0, pj)), f1 = lg_gsum(lg_dot(b1, pj)), f2 = lg_gsum(lg_dot(b2, pj)), f3 = lg_gsum(lg_dot(b3, pj));
                        const int e4 = l & 3;
                        const float hv2 = e4 == 0 ? q1.x : e4 == 1 ? q1.y : e4 == 2 ? q1.z : q1.w;
                        const float gv2 = e4 == 0 ? f0 : e4 == 1 ? f1 : e4 == 2 ? f2 : f3;
                        const uint32_t cv2 = (cd2 >> (8 * e4)) & 255u, sv2 = (sr2 >> (8 * e4)) & 255u;
                        if (l < 4 && hv2 != 0.0f && cv2) atomicAdd(&h_s[cv2], (double)hv2 * ((double)gv2 - (double)s_rd[sv2]));
                    }
                }
                if (fast) finish(rr, y);
            }
        }
        if (any_slow) {
#pragma unroll 1
            for (int u = 0; u < LG_U; ++u) {
                const int rr = g + LG_G * u;
                if (rr >= rows) break;
                const float4 r1 = s_rec[rr][1], rl = s_rec[rr][NREC - 1];
                const uint32_t srcs = __float_as_uint(TRANS ? rl.y : rl.z), meta = __float_as_uint(rl.w);
                if (!(meta & LG_SLOW)) continue;
                const int cnt_r = (int)((meta >> 16) & 255u), first_r = (int)(meta & 0xFFFFu);
                const float4 self = buf[rr][l];
                const float4 S = s_S[(meta >> 24) & 15u][l];
                int jn0; float w0; uint32_t cq;
                entry(first_r, !TRANS, jn0, w0, cq);                  // (transposed: the lists' LDS copy is gone)
                const int moff = cnt_r > 0 ? (int)(srcs & 255u) - jn0 : 0;        // first row of the molecule inside the block
                float4 y;
                if constexpr (!TRANS) {
                    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
                    for (int e = 0; e < cnt_r; ++e) {
                        int jn; float w;
                        entry(first_r + e, true, jn, w, cq);
                        lg_fma(acc, r1.y * (w - TINY), buf[min(moff + jn, LAGG_RB - 1)][l]);
                    }
                    const float wS = r1.y * TINY;
                    y.x = acc.x + fmaf(r1.x, self.x, wS * S.x);
                    y.y = acc.y + fmaf(r1.x, self.y, wS * S.y);
                    y.z = acc.z + fmaf(r1.x, self.z, wS * S.z);
                    y.w = acc.w + fmaf(r1.x, self.w, wS * S.w);
                } else {
                    // aggregation and edge gradients together
                    const float sj = rl.x, rs = r * sj;
                    const float4 pj = col_ok ? pj_load(u) : make_float4(0.f, 0.f, 0.f, 0.f);
                    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f), bs = acc;
                    bool self_bond = false;
                    for (int e = 0; e < cnt_r; ++e) {
                        int jn; float w; uint32_t c;
                        entry(first_r + e, false, jn, w, c);
                        const int src = min(moff + jn, LAGG_RB - 1);
                        const float sw = s_rs[src];
                        const float4 vv = buf[src][l];
                        lg_fma(acc, w * sw, vv);
                        lg_fma(bs, sw, vv);                           // (what the 1e-9 term must NOT count: s_i dY'_i of the bonded rows)
                        const float gd = lg_gsum(lg_dot(vv, pj));
                        if (l == 0 && sw != 0.0f) {
                            const float dU = sw * (gd - s_rd[src]);
                            if (c) atomicAdd(&h_s[c], (double)(dU * w * (1.0f - w)));
                            if (src == rr) dr_acc += (double)dU;      // (a self bond: the diagonal term below is NOT taken again)
                        }
                        self_bond = self_bond || src == rr;
                    }
                    y.x = acc.x + rs * self.x + TINY * (S.x - bs.x);
                    y.y = acc.y + rs * self.y + TINY * (S.y - bs.y);
                    y.z = acc.z + rs * self.z + TINY * (S.z - bs.z);
                    y.w = acc.w + rs * self.w + TINY * (S.w - bs.w);
                    const float gs = lg_gsum(lg_dot(self, pj));
                    if (l == 0 && sj != 0.0f && !self_bond) dr_acc += (double)(sj * (gs - s_rd[rr]));
                }
                finish(rr, y);
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
        }                                                             // (chunks)
    }                                                                 // (blocks)
    if constexpr (TRANS) {
        if (dr_acc != 0.0) atomicAdd(&h_s[256], dr_acc);
        __syncthreads();
        // non-zero bins -> one of the shared accumulator slabs (kernels.h EDGE_COPIES; drained by unpack_grads)
        double* out = ed.datt + ((size_t)((blockIdx.y + blockIdx.x) & (EDGE_COPIES - 1)) * ed.vc.K + k) * EDGE_SLAB;
        const double hv = h_s[tid];
        if (hv != 0.0) atomicAdd(&out[tid], hv);
        if (tid == 0 && h_s[256] != 0.0) atomicAdd(&out[256], h_s[256]);
    }
}

// ---- policy ------------------------------------------------------------------------------------------------------------------------------
// EAGCN_AGG = lds (always, N <= 256) | dense; default: by shape and DIRECTION.  Measured on MI355X, whole step in ms
// with this path forced in the backward only / in both directions against the matrix-core kernels of agg.hip (tools/r5_lagg_parts.sh,
// profiles/r05_lagg_policy.txt; the transposed kernel also absorbs bn_bwd_apply for the Concate layers):
//     K = 8, N = 256, all molecules 256 atoms, B = 1024 (BASELINE configs[4]):          - / 10.6      against 14.6
//     Tox21 (19 atoms on average, N = 132) B = 256, Concate (configs[1]):           0.442 / 0.428    against 0.447
//     Tox21 B = 1024:                                                               0.898 / 0.906    against 0.922
//     Lipo 3-layer Concate B = 512:                                                 1.350 / 1.357    against 1.401
//     HIV Weighted_sum (24 atoms on average, N = 222, 5 x 1250 columns), B = 1024:   7.19 / 7.15     against 6.94
// -> round-5 policy: forward for LARGE molecules (padded size from EAGCN_LAGG_MIN_N = 240 and -- once the rows the batches of the shape
//    really hold are known, eagcn_batch.t_hint -- 96 atoms per molecule on average) or batches of up to 256 molecules; backward: large
//    molecules, or any layer whose BatchNorm backward leaves its second pass to this kernel (Concate); Weighted_sum layers of small
//    molecules on the matrix cores.
// Round 6 (group sums through DPP, edge gradients inside the row loop, several chunks per workgroup, the Weighted_sum staging; same-box
// A/B against the round-5 kernels, tools/r6_ab_lib2.sh): HIV 6.90 -> 6.10 ms with BOTH directions here, Tox21 B = 1024 and Lipo equal
// or faster with the forward here as well -> every layer of every structure takes this path in both directions whenever the
// padded size allows it (N <= 256); agg.hip serves larger molecules, code-book relations and EAGCN_AGG=dense.
static int lagg_cpw(const eagcn_batch& bt, const ViewCols& vc, bool trans);
static int lagg_policy() {
    static const int v = [] {
        const char* e = getenv("EAGCN_AGG");
        if (e && !strcmp(e, "lds")) return 1;
        if (e && !strcmp(e, "dense")) return 0;
        return 2;                                     // by shape
    }();
    return v;
}
static int lagg_min_n() {
    static const int v = [] { const char* e = getenv("EAGCN_LAGG_MIN_N"); return e ? atoi(e) : 240; }();
    return v;
}
static int lagg_fwd_maxb() {
    // (round 6: no limit by default.  Up to round 5 batches of more than 256 small molecules took the matrix-core forward of agg.hip;
    //  measured again with this round's kernels the step is equal or faster on this path -- Tox21 B = 1024 0.828 -> 0.822 ms, Lipo
    //  1.254 -> 1.258 -- and the default forward no longer sums the 1e-9 filler products one by one in fp32)
    static const int v = [] { const char* e = getenv("EAGCN_LAGG_FWD_MAXB"); return e ? atoi(e) : 0x7FFFFFFF; }();
    return v;
}
bool lagg_wfuse() {
    static const bool on = [] { const char* v = getenv("EAGCN_LAGG_WFUSE"); return !(v && v[0] == '0'); }();
    return on;
}
// should the index of batches of this shape carry bond lists and row blocks?  structure: EAGCN_STRUCT_* of the layers, or -1 (not known)
bool lagg_wanted(int B, int N, int structure) {
    const int p = lagg_policy();
    if (p == 0 || N > LAGG_RB) return false;
    if (p == 1) return true;
    return N >= lagg_min_n() || B <= lagg_fwd_maxb() || structure != EAGCN_STRUCT_WEIGHTED || lagg_wfuse();
}
int lagg_parts() {                                    // (debug switch) bit 0: forward, bit 1: backward on this path
    static const int v = [] { const char* e = getenv("EAGCN_LAGG_PARTS"); return e ? atoi(e) : 3; }();
    return v;
}
// does this batch take the path?  dir 0: forward, 1: backward; absorbs_bn: the layer's BatchNorm backward would leave its second pass here
bool lagg_use(const eagcn_batch* b, int dir, bool absorbs_bn, const ViewCols* vc) {
    if (!(b->build_lists && b->blk && b->mol_info && b->row_ptr && b->col_ptr && b->nbr && b->tnbr && b->ecode && b->tcode && b->row_info))
        return false;
    const int p = lagg_policy();
    if (p == 0 || b->N > LAGG_RB || !(lagg_parts() & (dir ? 2 : 1))) return false;
    if (p == 1) return true;
    const bool large = b->N >= lagg_min_n() && !(b->t_hint > 0 && (long)b->t_hint < 96L * b->B);
    if (large) return true;
    // forward of small molecules: batches of up to 256 molecules, or (round 6) launches of so many workgroups that each takes several
    // chunks of a block and builds its records once (HIV widths: agg_wave 0.79 -> 0.70 ms)
    return dir ? absorbs_bn : (b->B <= lagg_fwd_maxb() || (vc && lagg_cpw(*b, *vc, false) > 1));
}
// Rows per block: always LAGG_RB.  Measured (EAGCN_LAGG_RB = 256 / 128 / 64 / 32, whole step in ms): Tox21 B = 256 0.423 / 0.431 / 0.458 /
// 0.541, B = 1024 0.889 / 0.906 / 0.991 / 1.172, Lipo B = 512 1.348 / 1.375 / 1.518 / 1.641 -- more, smaller blocks (more workgroups per CU
// to hide a block's barrier-separated phases) lose: a block's fixed cost (eight clamped row loads per thread whatever the row count, the
// record pass, four barriers) is what a launch pays per block, and the per-molecule column sums stop being shared.
int lagg_block_rows(const eagcn_batch* b) {
    static const int fixed = [] { const char* e = getenv("EAGCN_LAGG_RB"); return e ? atoi(e) : 0; }();
    (void)b;
    return fixed > 0 ? std::min(fixed, LAGG_RB) : LAGG_RB;
}
int lagg_slabs(const eagcn_batch* b) { return std::max(1, b->B); }

// chunks of the widest view / an ESTIMATE of the block count from the rows batches of this shape hold (a block closes at LAGG_RB rows or
// LAGG_MAXM molecules: 1.25 x the larger of the two quotients; the kernel loops, so any count is handled, and a tight grid spares the
// launch thousands of workgroups that would only find out that they have no block)
static void lagg_extent(const eagcn_batch& bt, const ViewCols& vc, int* chunks, int* gy) {
    int wmax = 0;
    for (int k = 0; k < vc.K; ++k) wmax = std::max(wmax, vc.off[k + 1] - vc.off[k]);
    *chunks = cdiv(wmax, LG_CW);
    const int rows = bt.t_hint > 0 ? std::min(bt.t_hint, bt.T) : bt.T;
    const int est = 5 * std::max(rows / lagg_block_rows(&bt), bt.B / LAGG_MAXM) / 4 + 2;
    *gy = std::max(1, std::min(std::min(bt.B, est), 65535));
}
// chunks per workgroup: 1 while the launch is a few rounds of the chip's ~768 resident workgroups (the step then wants as many short
// workgroups as it can get: configs[1], B = 1024 and Lipo measured 3-5 % SLOWER with more), several where there are thousands (C5:
// 41 k, HIV widths: 26 k / 10 k): forward up to 8 (C5 forward 681 -> 593 us), transposed up to 4 (C5 -2 %, HIV: no difference between 1, 4, 8)
static int lagg_cpw(const eagcn_batch& bt, const ViewCols& vc, bool trans) {
    static const int cpw_env = [] { const char* e = getenv("EAGCN_LAGG_CPW"); return e ? atoi(e) : 0; }();
    static const int cpw_bwd = [] { const char* e = getenv("EAGCN_LAGG_CPW_BWD"); return e ? atoi(e) : 1; }();
    int chunks, gy;
    lagg_extent(bt, vc, &chunks, &gy);
    const long wgs = (long)vc.K * chunks * gy;
    int per = (trans && !cpw_bwd) ? 1 : cpw_env > 0 ? cpw_env : (int)std::min<long>(trans ? 4 : 8, std::max<long>(1, wgs / 3072));
    return std::max(1, std::min(per, chunks));
}
static int lagg_grid(const AggArgs& a, dim3* grid, int* nchunk, int* cpw, bool trans) {
    int chunks, gy;
    lagg_extent(a.bt, a.vc, &chunks, &gy);
    *cpw = lagg_cpw(a.bt, a.vc, trans);
    *nchunk = cdiv(chunks, *cpw);
    *grid = dim3((unsigned)(a.vc.K * *nchunk), (unsigned)gy);
    return EAGCN_OK;
}

int launch_lagg_fwd(AggArgs a, hipStream_t s) {
    if (a.bt.B == 0 || a.bt.T == 0) return EAGCN_OK;
    dim3 grid;
    int rc = lagg_grid(a, &grid, &a.nchunk, &a.cpw, false);
    if (rc) return rc;
    ProfScope ps(PROF_AGG, s);
    EdgeArgs e;
    memset(&e, 0, sizeof(e));
    static const int dbgm = [] { const char* e = getenv("EAGCN_LAGG_DBG"); return e ? atoi(e) : 0; }();
    a.xcd = dbgm;
    if (a.cpw > 1) lagg_kernel<false, true><<<grid, 256, 0, s>>>(a, e);
    else lagg_kernel<false, false><<<grid, 256, 0, s>>>(a, e);
    EAGCN_LAUNCH_CHECK();
    return EAGCN_OK;
}

int launch_lagg_bwd(AggArgs a, const EdgeArgs& e, hipStream_t s) {
    if (a.bt.B == 0 || a.bt.T == 0) return EAGCN_OK;
    EAGCN_CHECK_ARG(e.atomic && e.datt, "lagg: the edge gradients leave through the shared accumulator slabs (EdgeArgs.atomic)");
    dim3 grid;
    int rc = lagg_grid(a, &grid, &a.nchunk, &a.cpw, true);
    if (rc) return rc;
    ProfScope ps(PROF_AGG, s);
    static const int dbgm = [] { const char* e = getenv("EAGCN_LAGG_DBG"); return e ? atoi(e) : 0; }();
    a.xcd = dbgm;
    EAGCN_CHECK_ARG(!a.w_aw || a.bn_tab, "lagg: the upstream-gradient form needs the BatchNorm tables");
    if (a.w_aw) {
        // (one instantiation: with one chunk per workgroup the records of a block are simply built in front of its only chunk; the
        //  single-chunk body with this form's staging needs 177 registers)
        lagg_kernel<true, true, true><<<grid, 256, 0, s>>>(a, e);
    } else {
        if (a.cpw > 1) lagg_kernel<true, true><<<grid, 256, 0, s>>>(a, e);
        else lagg_kernel<true, false><<<grid, 256, 0, s>>>(a, e);
    }
    EAGCN_LAUNCH_CHECK();
    return EAGCN_OK;
}

}  // namespace eagcn

extern "C" int eagcn_agg_wants_bond_lists(int B, int N) { return eagcn::lagg_wanted(B, N, -1) ? 1 : 0; }
extern "C" int eagcn_agg_wants_bond_lists_for(int B, int N, int structure) { return eagcn::lagg_wanted(B, N, structure) ? 1 : 0; }
