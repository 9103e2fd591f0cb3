// Edge-attention aggregation over BOND LISTS: the same operator as agg.hip, evaluated as what it is.
//
// Reference semantics (layers.py:82-92 with the masks of layers.py:294-304), per molecule b and view k:
//     U[i,j]  = sigmoid(w_k[type(i,j)]) adj[i,j] + sigmoid(self_r) m_i [i == j] + 1e-9 (1 - adj[i,j])
//     A^[i,j] = m_i U[i,j] / sum_j' U[i,j'] ,   Y'[i,:] = sum_j A^[i,j] P_k[j,:]
// U is NOT a dense matrix: it is sigma at the two to four bonds of an atom, sigma(self_r) on the diagonal and the constant
// 1e-9 everywhere else, so
//     sum_j U[i,j] P[j,:] = sum_{bonds (i,j)} sigma_ij P[j,:] + r m_i P[i,:] + 1e-9 ( S_b - sum_{bonds (i,j)} P[j,:] ),   S_b = sum_{j < nat} P[j,:]
// -- a gather of deg + 1 rows plus ONE rank-one term per molecule.  agg.hip multiplies the full nat x nat block on the fp32
// matrix cores: for a 256-atom molecule 99 % of its MFMA work is a multiplication by 1e-9 (the K = 8 / N = 256 config spent
// 8.6 of its 16.3 ms there: profiles/r04_c5_synth_kernel_trace.txt), and for wide layers over small molecules (HIV: 1264 columns per
// view) it streams P at 1 TB/s.  Round 2 tried this formulation and DROPPED the rank-one term (5e-7 relative: three parity
// cases failed by 1.0-1.3x); here it is exact: a wavefront owns whole molecules, sums the molecule's rows once (S_b, in a
// fixed order: deterministic), then walks its rows.
//
// One wavefront per (molecule, view, 256-column chunk): lane = four adjacent columns (16-byte accesses, 1 KB per row and wave).
// Bond lists come from the batch index (index.hip index_csr_kernel: row lists for the forward, column lists for the
// transposed aggregation of the backward; one 64-bit word per bond holds the bond-type code of every view).  The list data of a
// row are wave-uniform: scalar loads.  Rows are taken SAGG_R at a time so that their list entries and neighbour rows are in
// flight together (a row alone is a chain of three dependent round trips).
// Forward epilogue as in agg.hip: row scale m_i / rowsum_i saved for the backward, BatchNorm partial sums (fp64) per column,
// one slab per workgroup.  Transposed: dP[j,:] = sum_{bonds (i,j)} s_i sigma_ij dY'[i,:] + s_j r dY'[j,:] + 1e-9 (G_b - sum_{bonds} s_i
// dY'[i,:]), G_b = sum_i s_i dY'[i,:]; written as fp32 or straight as the bf16 operand planes of the plane GEMMs.
#include <stdlib.h>

#include <algorithm>

#include "common.h"
#include "kernels.h"

namespace eagcn {

constexpr int SAGG_R = 4;                    // rows in flight per wavefront
constexpr int SAGG_D = 4;                    // list entries of a row requested together
constexpr int SAGG_NMAX = 256;               // atoms per molecule whose row pointers are staged in LDS (launch_sagg checks N)
constexpr int SAGG_ECAP = 1024;              // list entries per molecule staged in LDS (the rest is read from global memory)

__device__ __forceinline__ float4 sagg_ld(const float* p, bool ok) { return ok ? *reinterpret_cast<const float4*>(p) : make_float4(0.f, 0.f, 0.f, 0.f); }
__device__ __forceinline__ void sagg_fma(float4& acc, float w, const float4& v) {
    acc.x = fmaf(w, v.x, acc.x); acc.y = fmaf(w, v.y, acc.y); acc.z = fmaf(w, v.z, acc.z); acc.w = fmaf(w, v.w, acc.w);
}
__device__ __forceinline__ void sagg_add(float4& acc, const float4& v) { acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w; }

// The lists of a molecule go through LDS: a row's {first entry, count} and its entries {neighbour atom, weight} are the same
// for every lane, and read from global memory they were a chain of three dependent round trips per row batch (first version:
// 8.9 ms at the K = 8 / N = 256 config against 8.6 ms for the dense kernels).  One coalesced pass per molecule loads them,
// looks the sigma of every entry up and, for the transposed form, multiplies in s_i = m_i / rowsum_i of the row the bond comes from.
template <bool TRANS>
__global__ __launch_bounds__(256) void sagg_kernel(AggArgs a) {
    __shared__ float sig_s[256];
    __shared__ int2 ptr_s[4][SAGG_NMAX];                             // per wave: {first entry relative to the molecule, count}
    __shared__ short nb_s[4][SAGG_ECAP];                             // neighbour atom of an entry
    __shared__ float w_s[4][SAGG_ECAP];                              // sigma (forward) / s_i sigma (transposed)
    __shared__ float sw_s[TRANS ? 4 : 1][TRANS ? SAGG_ECAP : 1];     // transposed: s_i of the entry's row
    __shared__ float rs_s[4][SAGG_NMAX];                             // forward: m_i; transposed: s_i
    __shared__ double st_s[TRANS ? 1 : 4][TRANS ? 1 : 64][8];       // forward: per wave and lane: sum y, sum y^2 of its four columns
    const eagcn_batch& bt = a.bt;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int k = blockIdx.y / a.nchunk, cc = blockIdx.y - k * a.nchunk;
    const int wk = a.vc.off[k + 1] - a.vc.off[k];                   // padded width of the view (a multiple of 16)
    const int cl = cc * 256 + 4 * lane;                             // this lane's first column inside the view
    const bool col_ok = cl < wk;
    const int c0 = a.vc.off[k] + cl;
    const int T = dev_rows(bt);
    const int nlog = dev_n(bt);
    const float r = a.rsig[k];
    const float* rsk = a.rscale + (size_t)k * bt.T;
    const int2* ptrs = reinterpret_cast<const int2*>(TRANS ? bt.col_ptr : bt.row_ptr);
    const int32_t* nbr = TRANS ? bt.tnbr : bt.nbr;
    const uint64_t* codes = TRANS ? bt.tcode : bt.ecode;
    double s1[4] = {0.0, 0.0, 0.0, 0.0}, s2[4] = {0.0, 0.0, 0.0, 0.0};
    sig_s[threadIdx.x] = a.sig[k * 256 + threadIdx.x];
    __syncthreads();

    for (int b = blockIdx.x * 4 + wave; b < bt.B; b += gridDim.x * 4) {
        const int4 mi = reinterpret_cast<const int4*>(bt.mol_info)[b];          // {nat, row0, edge0, bonds}
        const int n = mi.x, r0 = mi.y, e0m = mi.z, ne = mi.w;
        if (n <= 0 || T == 0) continue;
        const float* base = a.src + (size_t)r0 * a.lds + c0;
        // ---- lists of the molecule -> LDS (wave-private: no barrier, LDS operations of a wave complete in order) ----------------------
        __builtin_amdgcn_wave_barrier();
        for (int i = lane; i < n; i += 64) {
            const int2 p = ptrs[r0 + i];
            ptr_s[wave][i] = make_int2(p.x - e0m, p.y);
            rs_s[wave][i] = TRANS ? rsk[r0 + i] : bt.row_m[r0 + i];
        }
        for (int e = lane; e < min(ne, SAGG_ECAP); e += 64) {
            const int jn = nbr[e0m + e];
            const uint32_t c = (uint32_t)(codes[e0m + e] >> (8 * k)) & 255u;
            nb_s[wave][e] = (short)jn;
            if constexpr (TRANS) {
                const float si = rsk[r0 + jn];
                sw_s[wave][e] = si;
                w_s[wave][e] = sig_s[c] * si;
            } else {
                w_s[wave][e] = sig_s[c];
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        auto entry = [&](int el, int& jn, float& w, float& sw) __attribute__((always_inline)) {     // entry `el` of the molecule's lists
            if (el < SAGG_ECAP) {
                jn = nb_s[wave][el]; w = w_s[wave][el];
                if constexpr (TRANS) sw = sw_s[wave][el]; else sw = 1.0f;
            } else {                                                                                 // (beyond the staged part: global memory)
                jn = nbr[e0m + el];
                const uint32_t c = (uint32_t)(codes[e0m + el] >> (8 * k)) & 255u;
                sw = TRANS ? rsk[r0 + jn] : 1.0f;
                w = sig_s[c] * sw;
            }
        };
        // ---- S_b (forward) / G_b = sum_i s_i dY'_i (transposed): one pass over the molecule's rows, eight rows in flight -----------
        float4 S = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int i0 = 0; i0 < n; i0 += 8) {
            float4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = sagg_ld(base + (size_t)min(i0 + u, n - 1) * a.lds, col_ok && i0 + u < n);
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                if constexpr (TRANS) sagg_fma(S, rs_s[wave][min(i0 + u, n - 1)], v[u]); else sagg_add(S, v[u]);
            }
        }
        // ---- the rows, SAGG_R at a time ------------------------------------------------------------------------------------------
        for (int i0 = 0; i0 < n; i0 += SAGG_R) {
            int first[SAGG_R], cnt[SAGG_R];
            float mrow[SAGG_R];
            float4 self[SAGG_R], acc[SAGG_R], bs[SAGG_R];
            float wsum[SAGG_R];
            int cmax = 0;
#pragma unroll
            for (int q = 0; q < SAGG_R; ++q) {
                const int i = min(i0 + q, n - 1);
                const int2 p = ptr_s[wave][i];
                first[q] = __builtin_amdgcn_readfirstlane(p.x);
                cnt[q] = i0 + q < n ? __builtin_amdgcn_readfirstlane(p.y) : 0;
                cmax = max(cmax, cnt[q]);
                mrow[q] = rs_s[wave][i];               // forward: m_i (a row without bonds is masked); transposed: s_j (zero for masked rows)
                self[q] = sagg_ld(base + (size_t)i * a.lds, col_ok && i0 + q < n);
                acc[q] = make_float4(0.f, 0.f, 0.f, 0.f);
                bs[q] = make_float4(0.f, 0.f, 0.f, 0.f);
                wsum[q] = 0.0f;
            }
            for (int e0 = 0; e0 < cmax; e0 += SAGG_D) {
                float ww[SAGG_R][SAGG_D], sv[SAGG_R][SAGG_D];
                float4 vv[SAGG_R][SAGG_D];
#pragma unroll
                for (int q = 0; q < SAGG_R; ++q)
#pragma unroll
                    for (int d = 0; d < SAGG_D; ++d) {
                        const bool live = e0 + d < cnt[q];
                        int jn = 0;
                        ww[q][d] = 0.0f; sv[q][d] = 0.0f;
                        if (live) entry(first[q] + e0 + d, jn, ww[q][d], sv[q][d]);
                        vv[q][d] = sagg_ld(base + (size_t)jn * a.lds, col_ok && live);
                    }
#pragma unroll
                for (int q = 0; q < SAGG_R; ++q)
#pragma unroll
                    for (int d = 0; d < SAGG_D; ++d) {
                        sagg_fma(acc[q], ww[q][d], vv[q][d]);
                        if constexpr (TRANS) sagg_fma(bs[q], sv[q][d], vv[q][d]);    // (what the 1e-9 term must NOT count: s_i dY'_i of the bonded rows)
                        else { if (e0 + d < cnt[q]) sagg_add(bs[q], vv[q][d]); wsum[q] += ww[q][d]; }
                    }
            }
#pragma unroll
            for (int q = 0; q < SAGG_R; ++q) {
                if (i0 + q >= n) continue;
                const int row = r0 + i0 + q;
                float4 y;
                if constexpr (!TRANS) {
                    // rowsum_i = sum sigma + r m_i + 1e-9 (columns without a bond, padding included)
                    const float d = wsum[q] + r * mrow[q] + TINY * (float)(nlog - cnt[q]);
                    const float sc = mrow[q] > 0.0f ? 1.0f / d : 0.0f;
                    if (cc == 0 && lane == 0) a.rscale[(size_t)k * bt.T + row] = sc;
                    const float rm = r * mrow[q];
                    y.x = sc * (acc[q].x + rm * self[q].x + TINY * (S.x - bs[q].x));
                    y.y = sc * (acc[q].y + rm * self[q].y + TINY * (S.y - bs[q].y));
                    y.z = sc * (acc[q].z + rm * self[q].z + TINY * (S.z - bs[q].z));
                    y.w = sc * (acc[q].w + rm * self[q].w + TINY * (S.w - bs[q].w));
                    s1[0] += (double)y.x; s2[0] += (double)y.x * (double)y.x;
                    s1[1] += (double)y.y; s2[1] += (double)y.y * (double)y.y;
                    s1[2] += (double)y.z; s2[2] += (double)y.z * (double)y.z;
                    s1[3] += (double)y.w; s2[3] += (double)y.w * (double)y.w;
                } else {
                    const float rs = r * mrow[q];                                  // s_j r (mrow = s_j here)
                    y.x = acc[q].x + rs * self[q].x + TINY * (S.x - bs[q].x);
                    y.y = acc[q].y + rs * self[q].y + TINY * (S.y - bs[q].y);
                    y.z = acc[q].z + rs * self[q].z + TINY * (S.z - bs[q].z);
                    y.w = acc[q].w + rs * self[q].w + TINY * (S.w - bs[q].w);
                }
                if (col_ok) {
                    if (TRANS && a.planes.p) bx_store4(a.planes, row, c0, y);
                    else *reinterpret_cast<float4*>(a.dst + (size_t)row * a.ldd + c0) = y;
                }
            }
        }
    }
    if constexpr (!TRANS) {
        // BatchNorm partial sums of this workgroup's four waves (different molecules, same columns) -> slab[bx][column][2]
#pragma unroll
        for (int e = 0; e < 4; ++e) { st_s[wave][lane][2 * e] = s1[e]; st_s[wave][lane][2 * e + 1] = s2[e]; }
        __syncthreads();
        if (wave == 0 && col_ok) {
            const int fp = a.vc.off[a.vc.K];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const double t1 = (st_s[0][lane][2 * e] + st_s[1][lane][2 * e]) + (st_s[2][lane][2 * e] + st_s[3][lane][2 * e]);
                const double t2 = (st_s[0][lane][2 * e + 1] + st_s[1][lane][2 * e + 1]) + (st_s[2][lane][2 * e + 1] + st_s[3][lane][2 * e + 1]);
                *reinterpret_cast<double2*>(a.stats + ((size_t)blockIdx.x * fp + c0 + e) * 2) = make_double2(t1, t2);
            }
        }
    }
}

// When does a batch take this path?  Only on request (EAGCN_AGG=sparse).  Measured on MI355X (gpurun_out, round 4; ms per step,
// aggregation + edge gradients inside): K = 8 / N = 256 config 6.9 + 2.5 against 8.6 for the matrix-core kernels of agg.hip, HIV
// widths 3.5 + 0.8 against 2.4, Tox21 batch 1024 0.73 + 0.12 against 0.25.  It does 64x fewer multiply-adds at 256 atoms and
// still loses: a wavefront that owns a molecule walks its rows four at a time, every batch one full memory round trip behind the
// previous one (the neighbour rows of batch t + 1 are only requested when batch t has been stored), two waves per SIMD at 190
// registers -- latency-bound at 1 - 1.5 TB/s, like the first version that read the lists with scalar loads (8.9 ms).  What would
// have to change is in DESIGN.md (next steps); the dense kernels stay the default for every shape.
static int sagg_policy() {
    static const int v = [] {
        const char* e = getenv("EAGCN_AGG");
        if (e && !strcmp(e, "sparse")) return 1;
        return 0;
    }();
    return v;
}
bool sagg_wanted(int B, int N) {
    const int p = sagg_policy();
    return N <= SAGG_NMAX && p == 1;
}
bool sagg_use(const eagcn_batch* b) {
    return b->build_lists && b->mol_info && b->row_ptr && b->col_ptr && b->nbr && b->tnbr && b->ecode && b->tcode && sagg_wanted(b->B, b->N);
}
// workgroups along x = BatchNorm partial slabs (every one of them is written)
int sagg_grid_x(const eagcn_batch* b) { return std::max(1, std::min(cdiv(b->B, 4), agg_grid_x(b))); }

int launch_sagg(AggArgs a, bool trans, hipStream_t s) {
    if (a.bt.B == 0 || a.bt.T == 0) return EAGCN_OK;
    int wmax = 0;
    for (int k = 0; k < a.vc.K; ++k) wmax = std::max(wmax, a.vc.off[k + 1] - a.vc.off[k]);
    a.nchunk = cdiv(wmax, 256);
    dim3 grid(sagg_grid_x(&a.bt), a.vc.K * a.nchunk);
    ProfScope ps(PROF_AGG, s);
    if (trans) sagg_kernel<true><<<grid, 256, 0, s>>>(a); else sagg_kernel<false><<<grid, 256, 0, s>>>(a);
    EAGCN_LAUNCH_CHECK();
    return EAGCN_OK;
}

}  // namespace eagcn

/* 1 when batches of this shape take the bond-list aggregation (csrc/sagg.hip): the caller's index must then carry bond lists
 * (eagcn_batch.build_lists = 1 before eagcn_index_rows) */
extern "C" int eagcn_agg_wants_bond_lists(int B, int N) { return (eagcn::sagg_wanted(B, N) || eagcn::lagg_wanted(B, N, -1)) ? 1 : 0; }
extern "C" int eagcn_agg_wants_bond_lists_for(int B, int N, int structure) { return (eagcn::sagg_wanted(B, N) || eagcn::lagg_wanted(B, N, structure)) ? 1 : 0; }
