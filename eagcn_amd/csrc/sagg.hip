// EXPERIMENTAL, opt-in (EAGCN_AGG=sparse): multi-view edge-attention aggregation over the BOND LISTS of the batch index,
// forward and backward.  NOT the default path: dropping the 1e-9 filler from the product (below) is a 5e-7 relative
// change of the attention operator, which three ill-conditioned parity cases amplify to 1.0-1.3x their tolerance, and
// the exact variants that were tried (per-molecule column sums staged through LDS; 8-row bins with fp64 LDS sums)
// measured SLOWER than the dense matrix-core kernels of agg.hip at the Tox21 shape.  Measurements in DESIGN.md.
//
// Reference semantics (layers.py:82-92 with the masks of layers.py:294-304), per molecule b, view k and the slice
// P_k = X.W_k of the flat product (the reference computes (A.X).W, layers.py:39-40; re-associated):
//     U[i,j]  = sigmoid(w_k[type(i,j)]) * adj[i,j] + sigmoid(self_r) * m_i * [i==j] + 1e-9 * (1 - adj[i,j])
//     A^[i,j] = m_i * U[i,j] / sum_j' U[i,j']          (j' over all N padded columns)
//     Y'[i,:] = sum_j A^[i,j] P_k[j,:]
// U is a dense N x N matrix only formally: away from the bonds and the diagonal every entry is the 1e-9 filler that keeps
// the row sum of an all-zero row away from zero (layers.py:294).  The kernels evaluate
//     Y'[i,:] = sc_i * ( sum_{bonds (i,j)} sigma_ij P_k[j,:]  +  r m_i P_k[i,:] ),
//     1/sc_i  = sum_{bonds (i,j)} sigma_ij + r m_i + 1e-9 (N - deg_i)              (sc_i = 0 where m_i = 0)
// i.e. the row sum EXACTLY (filler included) and the product without the filler terms, whose total weight relative to
// the bonded terms is below N * 1e-9 / sigma ~ 5e-7 for the largest supported molecule (the dense kernels of agg.hip
// already drop the filler columns beyond nat[b]; north_star's tolerance is 1e-5).  What is left is a gather over the two
// to four bonds of an atom: nat^2 multiply-adds per column become deg + 1 -- 5x fewer for a typical 18-atom molecule,
// 64x for a 256-atom one -- and no per-molecule structure remains: the unit of work is a ROW.
//
// Work decomposition.  grid.y = (view, 256-column range of the view); a workgroup is (row lanes) x (four-column groups)
// with a thread -> column-group map that is fixed for its life, and walks a stride of packed rows.  Per row: one
// descriptor (first row of the molecule, list position, degree), the bond list, then the deg + 1 source rows --
// 16 adjacent lanes read 256 contiguous bytes of a row, neighbour rows are at most nat rows away (L1/L2) -- and one
// 16-byte store.  The BatchNorm partial sums (sum y, sum y^2, fp64) of a thread's four columns stay in registers for
// the whole kernel.  No LDS tile, no barrier in the row loop, no dependence on the molecule size.
//
// Backward (one kernel): the BatchNorm-backward affine dY' = sc (dH - c1 - xhat c2) is evaluated wherever dY' is needed
// (no separate pass that writes dY'), then with the lists of bonds INTO a row
//     dP[j,:]  = sum_{bonds (i,j)} sigma_ij sc_i dY'[i,:] + r sc_j dY'[j,:]
//     dU[i,j]  = sc_i ( <dY'[i,:], P[j,:]> - <dY'[i,:], Y'[i,:]> )      at the bonds and on the diagonal
//     d w_k[c] += dU[i,j] s (1-s)  at bonds of type c ;   d self_r_k += dU[i,i] r (1-r)      (SURVEY.md 8a closed form)
// Every sum over columns is linear, so the column ranges of different workgroups add up in the fp64 partial slabs.
#include <algorithm>

#include <stdlib.h>

#include "common.h"
#include "kernels.h"

namespace eagcn {

constexpr int SA_NCG = 64;       // four-column groups per workgroup (256 columns): at least 4 row lanes

__device__ __forceinline__ float4 f4zero() { return make_float4(0.f, 0.f, 0.f, 0.f); }
__device__ __forceinline__ void f4fma(float4& a, float w, const float4& v) {
    a.x = fmaf(w, v.x, a.x); a.y = fmaf(w, v.y, a.y); a.z = fmaf(w, v.z, a.z); a.w = fmaf(w, v.w, a.w);
}
__device__ __forceinline__ float f4dot(const float4& a, const float4& b) { return (a.x * b.x + a.y * b.y) + (a.z * b.z + a.w * b.w); }
__device__ __forceinline__ const float4& ldf4(const float* p) { return *reinterpret_cast<const float4*>(p); }

// column groups [g_lo, g_lo + ncg) of view k handled by split `sp` (even split of the view's groups)
__device__ __forceinline__ void sagg_cols(const ViewCols& vc, int k, int sp, int& off, int& g_lo, int& ncg) {
    off = vc.off[k];
    const int gv = (vc.off[k + 1] - off) >> 2;
    const int nsp = (gv + SA_NCG - 1) / SA_NCG;
    const int per = nsp > 0 ? (gv + nsp - 1) / nsp : 0;
    g_lo = sp * per;
    ncg = sp < nsp ? min(per, gv - g_lo) : 0;
}

// ---- forward ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void sagg_fwd_kernel(SAggFwd a) {
    __shared__ double st_c[SA_NCG * 4 * 2];
    __shared__ float sig_s[256];
    const eagcn_batch& bt = a.bt;
    const int tid = threadIdx.x;
    const int bx = blockIdx.x, k = blockIdx.y / a.nsplit, sp = blockIdx.y % a.nsplit;
    int off, g_lo, ncg;
    sagg_cols(a.vc, k, sp, off, g_lo, ncg);
    if (ncg == 0) return;                                          // this view has fewer column ranges than the widest
    const int fp = a.vc.off[a.vc.K];
    const int RL = 256 / ncg, cg = tid % ncg, rl = tid / ncg;
    const int col = off + 4 * (g_lo + cg);
    sig_s[tid] = a.sig[k * 256 + tid];
    for (int c = tid; c < SA_NCG * 8; c += 256) st_c[c] = 0.0;
    const float rself = a.rsig[k];
    const int T = dev_rows(bt);
    const int nlog = dev_n(bt);
    double s1[4] = {0.0, 0.0, 0.0, 0.0}, s2[4] = {0.0, 0.0, 0.0, 0.0};
    __syncthreads();
    if (rl < RL) {
        const float* P = a.P + col;
        float* Y = a.Y + col;
        for (int r = bx * RL + rl; r < T; r += gridDim.x * RL) {
            const int r0 = bt.row_info[4 * r + 3];
            const int2 rp = reinterpret_cast<const int2*>(bt.row_ptr)[r];
            const float rm = rself * bt.row_m[r];
            const float4 self = ldf4(P + (size_t)r * a.ld);
            float4 acc = make_float4(rm * self.x, rm * self.y, rm * self.z, rm * self.w);
            float sum = 0.0f;
            for (int e0 = 0; e0 < rp.y; e0 += 4) {                 // four neighbour rows in flight
                int j[4];
                uint64_t c[4];
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const bool ok = e0 + t < rp.y;
                    j[t] = ok ? bt.nbr[rp.x + e0 + t] : 0;
                    c[t] = ok ? bt.ecode[rp.x + e0 + t] : 0ull;
                }
                float4 v[4];
#pragma unroll
                for (int t = 0; t < 4; ++t) v[t] = e0 + t < rp.y ? ldf4(P + (size_t)(r0 + j[t]) * a.ld) : f4zero();
#pragma unroll
                for (int t = 0; t < 4; ++t)
                    if (e0 + t < rp.y) {
                        const float w = sig_s[(uint32_t)(c[t] >> (8 * k)) & 255u];
                        sum += w;
                        f4fma(acc, w, v[t]);
                    }
            }
            const float d = sum + rm + TINY * (float)(nlog - rp.y);
            const float sc = rm > 0.0f ? 1.0f / d : 0.0f;           // (sigmoid(self_r) > 0: rm > 0 <=> m_i = 1)
            const float4 y = make_float4(acc.x * sc, acc.y * sc, acc.z * sc, acc.w * sc);
            *reinterpret_cast<float4*>(Y + (size_t)r * a.ld) = y;
            if (cg == 0 && sp == 0) a.rscale[(size_t)k * bt.T + r] = sc;
            s1[0] += (double)y.x; s2[0] += (double)y.x * (double)y.x;
            s1[1] += (double)y.y; s2[1] += (double)y.y * (double)y.y;
            s1[2] += (double)y.z; s2[2] += (double)y.z * (double)y.z;
            s1[3] += (double)y.w; s2[3] += (double)y.w * (double)y.w;
        }
    }
    if (a.stats) {
        // per-workgroup partial BatchNorm sums -> slab[bx][column][2]: every column of a slab is written by exactly one
        // workgroup (this view, this column range), so slabs need no clearing
        if (rl < RL) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                atomicAdd(&st_c[(4 * cg + u) * 2 + 0], s1[u]);
                atomicAdd(&st_c[(4 * cg + u) * 2 + 1], s2[u]);
            }
        }
        __syncthreads();
        double* slab = a.stats + ((size_t)bx * fp + off + 4 * g_lo) * 2;
        for (int c = tid; c < ncg * 8; c += 256) slab[c] = st_c[c];
    }
}

// ---- backward --------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void sagg_bwd_kernel(SAggBwd a) {
    __shared__ float4 kst_s[5][SA_NCG];            // BatchNorm-backward coefficients of the column groups (dot-product phase)
    __shared__ double h_s[256];
    __shared__ double dr_s[16];
    __shared__ float sig_s[256];
    const eagcn_batch& bt = a.bt;
    const int tid = threadIdx.x;
    const int bx = blockIdx.x, k = blockIdx.y / a.nsplit, sp = blockIdx.y % a.nsplit;
    int off, g_lo, ncg;
    sagg_cols(a.vc, k, sp, off, g_lo, ncg);
    const int fp = a.fp;
    double* out = a.datt + ((size_t)(bx * a.nsplit + sp) * a.vc.K + k) * EDGE_SLAB;
    if (ncg == 0) {                                                // no columns: the slab still has to read as zero
        out[tid] = 0.0;
        if (tid == 0) out[256] = 0.0;
        return;
    }
    const int RL = 256 / ncg, cg = tid % ncg, rl = tid / ncg;
    const int col = off + 4 * (g_lo + cg);
    // dY' = bsc (dH - c1 - (Y' - bmu) biv c2): coefficients of this thread's four columns
    const float4 bsc = ldf4(a.bn + BN_SC * fp + col), bmu = ldf4(a.bn + BN_MU * fp + col), biv = ldf4(a.bn + BN_INV * fp + col);
    const float4 c1 = ldf4(a.cc + col), c2 = ldf4(a.cc + fp + col);
    sig_s[tid] = a.sig[k * 256 + tid];
    h_s[tid] = 0.0;
    if (rl == 0) { kst_s[0][cg] = bsc; kst_s[1][cg] = bmu; kst_s[2][cg] = biv; kst_s[3][cg] = c1; kst_s[4][cg] = c2; }
    const float rself = a.rsig[k];
    const int T = dev_rows(bt);
    const float* rsk = a.rscale + (size_t)k * bt.T;
    auto dyp = [](const float4& dh, const float4& y, const float4& sc, const float4& mu, const float4& iv, const float4& k1,
                  const float4& k2) __attribute__((always_inline)) {
        float4 d;
        d.x = sc.x * (dh.x - k1.x - (y.x - mu.x) * iv.x * k2.x);
        d.y = sc.y * (dh.y - k1.y - (y.y - mu.y) * iv.y * k2.y);
        d.z = sc.z * (dh.z - k1.z - (y.z - mu.z) * iv.z * k2.z);
        d.w = sc.w * (dh.w - k1.w - (y.w - mu.w) * iv.w * k2.w);
        return d;
    };
    __syncthreads();
    // ---- transposed aggregation: dP[j] from the bonds INTO row j
    if (rl < RL) {
        const float* dH = a.dH + col;
        const float* Yp = a.Y + col;
        float* dP = a.dP + col;
        for (int r = bx * RL + rl; r < T; r += gridDim.x * RL) {
            const int r0 = bt.row_info[4 * r + 3];
            const int2 cp = reinterpret_cast<const int2*>(bt.col_ptr)[r];
            const float ws = rself * rsk[r];
            const float4 self = dyp(ldf4(dH + (size_t)r * fp), ldf4(Yp + (size_t)r * fp), bsc, bmu, biv, c1, c2);
            float4 acc = make_float4(ws * self.x, ws * self.y, ws * self.z, ws * self.w);
            for (int e0 = 0; e0 < cp.y; e0 += 4) {
                int i[4];
                uint64_t c[4];
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const bool ok = e0 + t < cp.y;
                    i[t] = ok ? r0 + bt.tnbr[cp.x + e0 + t] : r;
                    c[t] = ok ? bt.tcode[cp.x + e0 + t] : 0ull;
                }
                float4 dh[4], yy[4];
                float w[4];
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const bool ok = e0 + t < cp.y;
                    dh[t] = ok ? ldf4(dH + (size_t)i[t] * fp) : f4zero();
                    yy[t] = ok ? ldf4(Yp + (size_t)i[t] * fp) : f4zero();
                    w[t] = ok ? rsk[i[t]] : 0.0f;
                }
#pragma unroll
                for (int t = 0; t < 4; ++t)
                    if (e0 + t < cp.y)
                        f4fma(acc, w[t] * sig_s[(uint32_t)(c[t] >> (8 * k)) & 255u], dyp(dh[t], yy[t], bsc, bmu, biv, c1, c2));
            }
            *reinterpret_cast<float4*>(dP + (size_t)r * fp) = acc;
        }
    }
    // ---- bond and diagonal dot products <dY'[i], P[j]>: one 16-lane group per row, the dY' slice kept in registers
    double dr_acc = 0.0;
    {
        const int grp = tid >> 4, sl = tid & 15;
        const size_t gcol = (size_t)off + 4 * g_lo;
        for (int r = bx * 16 + grp; r < T; r += gridDim.x * 16) {
            const float rs = rsk[r];
            if (rs == 0.0f) continue;                             // m_i == 0: no dependence on the parameters
            const int r0 = bt.row_info[4 * r + 3];
            const int2 rp = reinterpret_cast<const int2*>(bt.row_ptr)[r];
            float4 dv[4];
            float rd = 0.0f;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int g = sl + 16 * u;
                dv[u] = f4zero();
                if (g < ncg) {
                    const float4 y = ldf4(a.Y + (size_t)r * fp + gcol + 4 * g);
                    dv[u] = dyp(ldf4(a.dH + (size_t)r * fp + gcol + 4 * g), y, kst_s[0][g], kst_s[1][g], kst_s[2][g], kst_s[3][g],
                                kst_s[4][g]);
                    rd += f4dot(dv[u], y);
                }
            }
#pragma unroll
            for (int o = 8; o > 0; o >>= 1) rd += __shfl_xor(rd, o);
            for (int e = -1; e < rp.y; ++e) {
                int j = r;
                uint32_t c = 0u;
                if (e >= 0) {
                    j = r0 + bt.nbr[rp.x + e];
                    c = (uint32_t)(bt.ecode[rp.x + e] >> (8 * k)) & 255u;
                }
                float g = 0.0f;
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    if (sl + 16 * u < ncg) g += f4dot(dv[u], ldf4(a.P + (size_t)j * fp + gcol + 4 * (sl + 16 * u)));
#pragma unroll
                for (int o = 8; o > 0; o >>= 1) g += __shfl_xor(g, o);
                if (sl == 0) {
                    const float dU = rs * (g - rd);
                    if (e < 0) dr_acc += (double)dU;
                    else if (c) {
                        const float sg = sig_s[c];
                        atomicAdd(&h_s[c], (double)(dU * sg * (1.0f - sg)));
                    }
                }
            }
        }
    }
    if ((tid & 15) == 0) dr_s[tid >> 4] = dr_acc;
    __syncthreads();
    // slab[workgroup][k][0..255] = bond-type histogram, [256] = self term (layout of unpack_grads_kernel, layer.hip)
    out[tid] = h_s[tid];
    if (tid == 0) {
        double t = 0.0;
        for (int q = 0; q < 16; ++q) t += dr_s[q];
        out[256] = t;
    }
}

// ---- host side ---------------------------------------------------------------------------------------------------------
bool sagg_enabled() {
    static const bool on = [] { const char* v = getenv("EAGCN_AGG"); return v && v[0] == 's'; }();      // EAGCN_AGG=sparse (default: agg.hip)
    return on;
}
// workgroups along x (= BatchNorm / edge-gradient partial slabs): enough rows per workgroup to amortise its prologue
int sagg_grid_x(const eagcn_batch* b) {
    static const int rows_per_wg = [] { const char* v = getenv("EAGCN_SAGG_ROWS"); return v ? std::max(1, atoi(v)) : 16; }();
    return std::max(1, std::min(cdiv(std::max(b->T, 1), rows_per_wg), 512));
}
// column ranges (workgroups) per view: ceil(widest view / 256 columns)
int sagg_nsplit(const eagcn_batch* b, int fmax) { (void)b; return std::max(1, cdiv(fmax / 4, SA_NCG)); }

static int sagg_check(const eagcn_batch& bt, const ViewCols& vc) {
    EAGCN_CHECK_ARG(bt.row_info && bt.row_ptr && bt.col_ptr, "aggregation: the batch index has no bond lists");
    for (int k = 0; k < vc.K; ++k) EAGCN_CHECK_ARG(((vc.off[k + 1] - vc.off[k]) & 15) == 0, "aggregation: view widths must be padded to 16");
    return EAGCN_OK;
}

int launch_sagg_fwd(const SAggFwd& a, hipStream_t s) {
    int rc = sagg_check(a.bt, a.vc);
    if (rc) return rc;
    dim3 grid(sagg_grid_x(&a.bt), a.vc.K * a.nsplit);
    ProfScope ps(PROF_AGG, s);
    sagg_fwd_kernel<<<grid, 256, 0, s>>>(a);
    EAGCN_LAUNCH_CHECK();
    return EAGCN_OK;
}

int launch_sagg_bwd(const SAggBwd& a, hipStream_t s) {
    int rc = sagg_check(a.bt, a.vc);
    if (rc) return rc;
    dim3 grid(sagg_grid_x(&a.bt), a.vc.K * a.nsplit);
    ProfScope ps(PROF_AGG, s);
    sagg_bwd_kernel<<<grid, 256, 0, s>>>(a);
    EAGCN_LAUNCH_CHECK();
    return EAGCN_OK;
}

}  // namespace eagcn
