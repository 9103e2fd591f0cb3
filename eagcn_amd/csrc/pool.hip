// Diff_Pooling read-out, molfp_mode='pool' (reference layers.py:492-506, models.py:90-92, 104-106):
//     A   = the last layer's attention matrix                       [B,N,N]   (layers.py:319-324 / :250-253 / :189)
//     AX  = A . x2                                                    [B,N,F]   (layers.py:39, bmm)
//     X'  = relu(AX . Wf)          S = softmax_p(AX . Ws)             [B,N,F] / [B,N,P]   (layers.py:499-500)
//     Pm  = S^T . X'               g = sum_p relu(Pm)                 [B,P,F] / [B,F]     (layers.py:501-503, models.py:106)
// Both products with weights are ONE flat GEMM (columns [Wf | Ws]) on the fp32 MFMA kernel of gemm.hip, issued by the host side
// between the entry points below; this file holds the per-molecule pieces and their backward passes.  The pooled adjacency
// S^T.A.S is returned by the reference's module but models.py never reads it: not computed.
//
// Only stored rows exist: a non-stored row (i >= nat[b]) has A row 0 (layers.py:324 multiplies by the row mask), hence AX = 0,
// X' = relu(0) = 0 (both bases are bias-free, layers.py:21) and it adds nothing to Pm.  As COLUMNS the non-stored rows do take
// part in A.x2 with the 1e-9 filler weight: they all hold the same vector pad_row, so their share is padsum[r] * pad_row.
//
// A is kept as packed rows [T][lda] (one row per stored atom, lda >= N).  Not a throughput path: plain FMA kernels, one
// wavefront per row or per 64 columns, no MFMA.
#include <algorithm>

#include "common.h"
#include "kernels.h"

namespace eagcn {

constexpr int POOL_MAXP = EAGCN_POOL_MAX;
constexpr int POOL_ACC = EAGCN_MAX_VIEWS * 256 + EAGCN_MAX_VIEWS + 1;   // [k][code] | dave_a[k] | dself_r

struct PoolAtt {
    int K, mode;
    int att_c[EAGCN_MAX_VIEWS];
    const float* att_w[EAGCN_MAX_VIEWS];
    const float* ave_a;
    const float* self_r;
};
static PoolAtt pool_att(const eagcn_pool_att* p) {
    PoolAtt a;
    a.K = p->K;
    a.mode = p->mode;
    for (int k = 0; k < EAGCN_MAX_VIEWS; ++k) {
        a.att_c[k] = k < p->K ? p->att_c[k] : 0;
        a.att_w[k] = k < p->K ? p->att_w[k] : nullptr;
    }
    a.ave_a = p->ave_a;
    a.self_r = p->self_r;
    return a;
}

__device__ __forceinline__ int pool_exact_to_packed(const ColMapD& m, int ce) {
    int eo = 0, po = 0;
    for (int s = 0; s < m.nseg; ++s) {
        if (ce < eo + m.w[s]) return po + (ce - eo);
        eo += m.w[s];
        po += m.p[s];
    }
    return 0;
}
__device__ __forceinline__ int pool_packed_to_exact(const ColMapD& m, int cp) {
    int eo = 0, po = 0;
    for (int s = 0; s < m.nseg; ++s) {
        if (cp < po + m.p[s]) return (cp - po < m.w[s]) ? eo + (cp - po) : -1;
        eo += m.w[s];
        po += m.p[s];
    }
    return -1;
}

// sigma(att_k . relation vector) of bond code c (1-based) in view k; 0 for a code outside the view's range
__device__ __forceinline__ float pool_view_sig(const eagcn_batch& bt, const PoolAtt& pa, int k, int c) {
    if (c < 1 || c > bt.channels[k]) return 0.0f;
    float s;
    if (!bt.rel_vec[k]) {
        if (c > pa.att_c[k]) return 0.0f;
        s = pa.att_w[k][c - 1];
    } else {
        const float* v = bt.rel_vec[k] + (size_t)(c - 1) * bt.rel_c[k];
        s = 0.0f;
        for (int ch = 0; ch < bt.rel_c[k]; ++ch) s = fmaf(pa.att_w[k][ch], v[ch], s);
    }
    return sigmoidf_(s);
}
__device__ __forceinline__ int pool_code(const eagcn_batch& bt, int k, int b, int i, int j) {
    return (int)bt.code[(((size_t)k * bt.B + b) * bt.N + i) * bt.ldc + j];
}

// ---- A: one wavefront per stored row ------------------------------------------------------------------------------------
// mode 0 (layers.py:319-324): u = sigma(sum_k aveA_k sigma(att_k[type_k])) at bonds, + sigma(self_r) on the diagonal of rows
//         with a bond, 1e-9 where there is no bond; A = u / rowsum(u), rows without a bond zeroed
// mode 1 (layers.py:250-253): the same with both sigmoids replaced by 1
// mode 2 (layers.py:189):     A = adj + mask * I, neither filler nor normalisation
__global__ __launch_bounds__(256) void pool_att_fwd_kernel(eagcn_batch bt, PoolAtt pa, float* __restrict__ A, int lda,
                                                            float* __restrict__ rinv, float* __restrict__ padsum) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = blockIdx.x * 4 + wave;
    if (r >= dev_rows(bt)) return;
    const int4 info = reinterpret_cast<const int4*>(bt.row_info)[r];          // {molecule, atom, nat, row0}
    const int b = info.x, i = info.y, nat = info.z;
    const int N = dev_n(bt);
    float* Ar = A + (size_t)r * lda;
    if (bt.row_m[r] == 0.0f) {
        for (int j = lane; j < N; j += 64) Ar[j] = 0.0f;
        if (lane == 0) { rinv[r] = 0.0f; padsum[r] = 0.0f; }
        return;
    }
    const float diag = pa.mode == 0 ? sigmoidf_(pa.self_r[0]) : 1.0f;
    const float fill = pa.mode == 2 ? 0.0f : TINY;
    const uint8_t* c0row = bt.code + ((size_t)b * bt.N + i) * bt.ldc;         // view 0: every view has the same bond positions
    float sum = 0.0f, tail = 0.0f;
    for (int j = lane; j < N; j += 64) {
        float u = fill;
        if (c0row[j]) {
            u = 1.0f;
            if (pa.mode == 0) {
                float a = 0.0f;
                for (int k = 0; k < pa.K; ++k) a += pa.ave_a[k] * pool_view_sig(bt, pa, k, pool_code(bt, k, b, i, j));
                u = sigmoidf_(a);
            }
        }
        if (j == i) u += diag;
        Ar[j] = u;
        sum += u;
        if (j >= nat) tail += u;
    }
    sum = wave_sum(sum);
    tail = wave_sum(tail);
    if (pa.mode != 2) {
        for (int j = lane; j < N; j += 64) Ar[j] = Ar[j] / sum;                // each lane re-reads what it wrote
        tail = tail / sum;
    }
    if (lane == 0) {
        rinv[r] = pa.mode != 2 ? 1.0f / sum : 1.0f;
        padsum[r] = tail;
    }
}

// dA -> d att_k[c], d aveA_k, d self_r (mode 0).  With s_i = rowsum(u): dL/du_ij = (dA_ij - sum_j' dA_ij' A_ij') / s_i.
// Sums go through fp64 LDS tables per workgroup, then fp64 atomics into acc[POOL_ACC]; pool_att_bwd_final converts.
__global__ __launch_bounds__(256) void pool_att_bwd_kernel(eagcn_batch bt, PoolAtt pa, const float* __restrict__ A, int lda,
                                                            const float* __restrict__ rinv, const float* __restrict__ dA,
                                                            double* __restrict__ acc) {
    __shared__ double l_acc[POOL_ACC];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int K = pa.K;
    const int n_acc = K * 256 + K + 1;
    for (int e = threadIdx.x; e < n_acc; e += 256) l_acc[e] = 0.0;
    __syncthreads();
    const int rows = dev_rows(bt), N = dev_n(bt);
    const float sr = sigmoidf_(pa.self_r[0]);
    for (int r = blockIdx.x * 4 + wave; r < rows; r += gridDim.x * 4) {
        if (bt.row_m[r] == 0.0f) continue;
        const int4 info = reinterpret_cast<const int4*>(bt.row_info)[r];
        const int b = info.x, i = info.y;
        const float* Ar = A + (size_t)r * lda;
        const float* dAr = dA + (size_t)r * lda;
        float dot = 0.0f;
        for (int j = lane; j < N; j += 64) dot += dAr[j] * Ar[j];
        dot = wave_sum(dot);
        const float inv = rinv[r];
        const uint8_t* c0row = bt.code + ((size_t)b * bt.N + i) * bt.ldc;
        for (int j = lane; j < N; j += 64) {
            const float du = (dAr[j] - dot) * inv;
            if (j == i) atomicAdd(&l_acc[K * 256 + K], (double)(du * sr * (1.0f - sr)));
            if (c0row[j]) {
                float a = 0.0f;
                for (int k = 0; k < K; ++k) a += pa.ave_a[k] * pool_view_sig(bt, pa, k, pool_code(bt, k, b, i, j));
                const float sa = sigmoidf_(a);
                const float da = du * sa * (1.0f - sa);
                for (int k = 0; k < K; ++k) {
                    const int c = pool_code(bt, k, b, i, j);
                    const float s = pool_view_sig(bt, pa, k, c);
                    if (s == 0.0f) continue;
                    atomicAdd(&l_acc[K * 256 + k], (double)(da * s));
                    atomicAdd(&l_acc[k * 256 + c], (double)(da * pa.ave_a[k] * s * (1.0f - s)));
                }
            }
        }
    }
    __syncthreads();
    for (int e = threadIdx.x; e < n_acc; e += 256)
        if (l_acc[e] != 0.0) atomicAdd(&acc[e], l_acc[e]);
}
struct PoolAttGrad {
    float* datt_w[EAGCN_MAX_VIEWS];
    float* dave_a;
    float* dself_r;
};
__global__ __launch_bounds__(256) void pool_att_bwd_final_kernel(eagcn_batch bt, PoolAtt pa, const double* __restrict__ acc,
                                                                  PoolAttGrad g) {
    const int K = pa.K;
    for (int e = threadIdx.x; e < K * 256; e += 256) {
        const int k = e >> 8, ch = e & 255;
        if (ch >= pa.att_c[k] || !g.datt_w[k]) continue;
        double t = 0.0;
        if (!bt.rel_vec[k]) {
            t = ch + 1 <= bt.channels[k] ? acc[k * 256 + ch + 1] : 0.0;
        } else {                                            // logit = <att_w, vec[c]>: d att_w[ch] = sum_c d logit_c vec[c][ch]
            for (int c = 1; c <= bt.channels[k]; ++c)
                t += acc[k * 256 + c] * (double)bt.rel_vec[k][(size_t)(c - 1) * bt.rel_c[k] + ch];
        }
        g.datt_w[k][ch] = (float)t;
    }
    if ((int)threadIdx.x < K && g.dave_a) g.dave_a[threadIdx.x] = (float)acc[K * 256 + threadIdx.x];
    if (threadIdx.x == 0 && g.dself_r) g.dself_r[0] = (float)acc[K * 256 + K];
}

// ---- AX = A . x2 over the stored rows: grid (B, ceil(F/64)), 4 wavefronts split the molecule's rows, lanes own columns ---
__global__ __launch_bounds__(256) void pool_mix_fwd_kernel(eagcn_batch bt, const float* __restrict__ A, int lda,
                                                            const float* __restrict__ padsum, const float* __restrict__ x,
                                                            ColMapD m, int ld, const float* __restrict__ pad_row,
                                                            float* __restrict__ AX, int F) {
    const int b = blockIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int f = blockIdx.y * 64 + lane;
    if (f >= F) return;
    const int nat = bt.nat[b], r0 = bt.row0[b];
    const int cp = pool_exact_to_packed(m, f);
    const float pv = pad_row ? pad_row[cp] : 0.0f;
    for (int i = wave; i < nat; i += 4) {
        const float* Ar = A + (size_t)(r0 + i) * lda;
        float acc = 0.0f;
        for (int j0 = 0; j0 < nat; j0 += 4) {
            float a[4], v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int j = j0 + u;
                a[u] = j < nat ? Ar[j] : 0.0f;
                v[u] = j < nat ? x[(size_t)(r0 + j) * ld + cp] : 0.0f;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) acc = fmaf(a[u], v[u], acc);
        }
        acc = fmaf(padsum[r0 + i], pv, acc);
        AX[(size_t)(r0 + i) * F + f] = acc;
    }
}
// dx[j] = sum_i A[i][j] dAX[i]  (exact columns only; the caller zero-fills dx so that padded columns stay 0)
__global__ __launch_bounds__(256) void pool_mix_bwd_x_kernel(eagcn_batch bt, const float* __restrict__ A, int lda,
                                                              const float* __restrict__ dAX, ColMapD m, int ld,
                                                              float* __restrict__ dx, int F) {
    const int b = blockIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int f = blockIdx.y * 64 + lane;
    if (f >= F) return;
    const int nat = bt.nat[b], r0 = bt.row0[b];
    const int cp = pool_exact_to_packed(m, f);
    for (int j = wave; j < nat; j += 4) {
        float acc = 0.0f;
        for (int i0 = 0; i0 < nat; i0 += 4) {
            float a[4], v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int i = i0 + u;
                a[u] = i < nat ? A[(size_t)(r0 + i) * lda + j] : 0.0f;
                v[u] = i < nat ? dAX[(size_t)(r0 + i) * F + f] : 0.0f;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) acc = fmaf(a[u], v[u], acc);
        }
        dx[(size_t)(r0 + j) * ld + cp] = acc;
    }
}
// dA[r][j] = <dAX[r], X_j>, X_j = x row of atom j, or pad_row for the non-stored columns.  One wavefront per row.
__global__ __launch_bounds__(256) void pool_mix_bwd_a_kernel(eagcn_batch bt, const float* __restrict__ x, ColMapD m, int ld,
                                                              const float* __restrict__ pad_row,
                                                              const float* __restrict__ dAX, int F,
                                                              float* __restrict__ dA, int lda) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = blockIdx.x * 4 + wave;
    if (r >= dev_rows(bt)) return;
    const int4 info = reinterpret_cast<const int4*>(bt.row_info)[r];
    const int nat = info.z, r0 = info.w;
    const int N = dev_n(bt);
    const float* d = dAX + (size_t)r * F;
    float* dAr = dA + (size_t)r * lda;
    for (int j = 0; j < nat; ++j) {
        const float* xr = x + (size_t)(r0 + j) * ld;
        float acc = 0.0f;
        for (int f = lane; f < F; f += 64) acc = fmaf(d[f], xr[pool_exact_to_packed(m, f)], acc);
        acc = wave_sum(acc);
        if (lane == 0) dAr[j] = acc;
    }
    float t = 0.0f;
    if (pad_row) {
        for (int f = lane; f < F; f += 64) t = fmaf(d[f], pad_row[pool_exact_to_packed(m, f)], t);
        t = wave_sum(t);
    }
    for (int j = nat + lane; j < N; j += 64) dAr[j] = t;
}
// d pad_row[cp] = sum_r padsum[r] dAX[r][exact(cp)]
__global__ __launch_bounds__(256) void pool_mix_bwd_pad_kernel(eagcn_batch bt, const float* __restrict__ padsum,
                                                                const float* __restrict__ dAX, ColMapD m, int ld, int F,
                                                                float* __restrict__ dpad) {
    const int cp = blockIdx.x * blockDim.x + threadIdx.x;
    if (cp >= ld) return;
    const int ce = pool_packed_to_exact(m, cp);
    double acc = 0.0;
    if (ce >= 0) {
        const int rows = dev_rows(bt);
        for (int r = 0; r < rows; ++r) acc += (double)(padsum[r] * dAX[(size_t)r * F + ce]);
    }
    dpad[cp] = (float)acc;
}

// ---- S = softmax over the P assignment logits Z[r][F .. F+P) -------------------------------------------------------------
__global__ __launch_bounds__(256) void pool_softmax_kernel(eagcn_batch bt, const float* __restrict__ Z, int ldz, int F, int P,
                                                            float* __restrict__ S) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= dev_rows(bt)) return;
    const float* z = Z + (size_t)r * ldz + F;
    float mx = z[0];
    for (int p = 1; p < P; ++p) mx = fmaxf(mx, z[p]);
    float sum = 0.0f;
    for (int p = 0; p < P; ++p) sum += expf(z[p] - mx);
    for (int p = 0; p < P; ++p) S[(size_t)r * P + p] = expf(z[p] - mx) / sum;
}
// Pm[b][p][f] = sum_i S[i][p] relu(Z[i][f]);  g[b][f] = sum_p relu(Pm[b][p][f]).  grid (B, ceil(F/64)), 4 wavefronts split rows
__global__ __launch_bounds__(256) void pool_reduce_fwd_kernel(eagcn_batch bt, const float* __restrict__ Z, int ldz, int F,
                                                               int P, const float* __restrict__ S, float* __restrict__ Pm,
                                                               float* __restrict__ g) {
    __shared__ float part[4][POOL_MAXP][64];
    const int b = blockIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int f = blockIdx.y * 64 + lane;
    const bool ok = f < F;
    const int nat = bt.nat[b], r0 = bt.row0[b];
    float acc[POOL_MAXP];
#pragma unroll
    for (int p = 0; p < POOL_MAXP; ++p) acc[p] = 0.0f;
    for (int i = wave; i < nat; i += 4) {
        const int r = r0 + i;
        const float xf = ok ? fmaxf(Z[(size_t)r * ldz + f], 0.0f) : 0.0f;
#pragma unroll
        for (int p = 0; p < POOL_MAXP; ++p)
            if (p < P) acc[p] = fmaf(S[(size_t)r * P + p], xf, acc[p]);
    }
#pragma unroll
    for (int p = 0; p < POOL_MAXP; ++p) part[wave][p][lane] = acc[p];
    __syncthreads();
    if (wave == 0 && ok) {
        float gs = 0.0f;
#pragma unroll
        for (int p = 0; p < POOL_MAXP; ++p)
            if (p < P) {
                const float t = (part[0][p][lane] + part[1][p][lane]) + (part[2][p][lane] + part[3][p][lane]);
                Pm[((size_t)b * P + p) * F + f] = t;
                gs += fmaxf(t, 0.0f);
            }
        g[(size_t)b * F + f] = gs;
    }
}
// dZ of one stored row per wavefront: dPm[p][f] = dg[f] [Pm > 0];  dZ[f] = [Z > 0] sum_p S_p dPm[p][f];
// dS_p = sum_f relu(Z[f]) dPm[p][f];  dZ[F+p] = S_p (dS_p - sum_q S_q dS_q)
__global__ __launch_bounds__(256) void pool_reduce_bwd_kernel(eagcn_batch bt, const float* __restrict__ Z, int ldz, int F,
                                                               int P, const float* __restrict__ S,
                                                               const float* __restrict__ Pm, const float* __restrict__ dg,
                                                               float* __restrict__ dZ) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = blockIdx.x * 4 + wave;
    if (r >= dev_rows(bt)) return;
    const int b = bt.row_mol[r];
    float sp[POOL_MAXP], ds[POOL_MAXP];
#pragma unroll
    for (int p = 0; p < POOL_MAXP; ++p) {
        sp[p] = p < P ? S[(size_t)r * P + p] : 0.0f;
        ds[p] = 0.0f;
    }
    for (int f = lane; f < F; f += 64) {
        const float z = Z[(size_t)r * ldz + f];
        const float xf = fmaxf(z, 0.0f);
        const float dgf = dg[(size_t)b * F + f];
        float dxf = 0.0f;
#pragma unroll
        for (int p = 0; p < POOL_MAXP; ++p)
            if (p < P) {
                const float dp = Pm[((size_t)b * P + p) * F + f] > 0.0f ? dgf : 0.0f;
                dxf = fmaf(sp[p], dp, dxf);
                ds[p] = fmaf(xf, dp, ds[p]);
            }
        dZ[(size_t)r * ldz + f] = z > 0.0f ? dxf : 0.0f;
    }
    float dotp = 0.0f;
#pragma unroll
    for (int p = 0; p < POOL_MAXP; ++p) {
        ds[p] = wave_sum(ds[p]);
        dotp = fmaf(sp[p], ds[p], dotp);
    }
#pragma unroll
    for (int p = 0; p < POOL_MAXP; ++p)
        if (p < P && lane == p) dZ[(size_t)r * ldz + F + p] = sp[p] * (ds[p] - dotp);
    for (int c = F + P + lane; c < ldz; c += 64) dZ[(size_t)r * ldz + c] = 0.0f;
}

static int rows_grid(const eagcn_batch* b) { return cdiv(std::max(b->T, 1), 4); }

}  // namespace eagcn

using namespace eagcn;

extern "C" size_t eagcn_pool_scratch_bytes(void) { return align256(POOL_ACC * sizeof(double)); }

extern "C" int eagcn_pool_attention_forward(const eagcn_batch* b, const eagcn_pool_att* p, float* A, int lda, float* rinv,
                                            float* padsum, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    EAGCN_CHECK_ARG(b && p, "eagcn_pool_attention_forward: null argument");
    EAGCN_CHECK_ARG(b->T == 0 || (A && rinv && padsum), "eagcn_pool_attention_forward: null output");
    EAGCN_CHECK_ARG(p->mode >= 0 && p->mode <= 2, "eagcn_pool_attention_forward: mode %d", p->mode);
    EAGCN_CHECK_ARG(lda >= b->N, "eagcn_pool_attention_forward: lda %d < N %d", lda, b->N);
    if (p->mode == 0) {
        EAGCN_CHECK_ARG(p->K == b->K && p->K >= 1 && p->K <= EAGCN_MAX_VIEWS, "eagcn_pool_attention_forward: %d views, the batch %d",
                        p->K, b->K);
        EAGCN_CHECK_ARG(p->ave_a && p->self_r, "eagcn_pool_attention_forward: null ave_A / self_r");
        for (int k = 0; k < p->K; ++k) EAGCN_CHECK_ARG(p->att_w[k], "eagcn_pool_attention_forward: null att_w[%d]", k);
    }
    ProfScope ps(PROF_READOUT, s);
    pool_att_fwd_kernel<<<rows_grid(b), 256, 0, s>>>(*b, pool_att(p), A, lda, rinv, padsum);
    EAGCN_LAUNCH_CHECK();
    return EAGCN_OK;
}

extern "C" int eagcn_pool_attention_backward(const eagcn_batch* b, const eagcn_pool_att* p, const float* A, int lda,
                                             const float* rinv, const float* dA, void* scratch, size_t scratch_bytes,
                                             void* stream) {
    hipStream_t s = (hipStream_t)stream;
    EAGCN_CHECK_ARG(b && p && scratch, "eagcn_pool_attention_backward: null argument");
    EAGCN_CHECK_ARG(p->mode == 0, "eagcn_pool_attention_backward: only mode 0 has parameters");
    EAGCN_CHECK_ARG(p->K == b->K && p->K >= 1 && p->K <= EAGCN_MAX_VIEWS, "eagcn_pool_attention_backward: %d views, the batch %d",
                    p->K, b->K);
    EAGCN_CHECK_ARG(b->T == 0 || (A && rinv && dA), "eagcn_pool_attention_backward: null input");
    if (scratch_bytes < eagcn_pool_scratch_bytes()) {
        set_error("eagcn_pool_attention_backward: scratch %zu < %zu bytes", scratch_bytes, eagcn_pool_scratch_bytes());
        return EAGCN_ERR_SCRATCH;
    }
    ProfScope ps(PROF_READOUT, s);
    double* acc = (double*)scratch;
    { int rcz = zero_fill(acc, POOL_ACC * sizeof(double), s); if (rcz) return rcz; }       // (a kernel: see kernels.h zero_fill)
    const int grid = std::max(1, std::min(rows_grid(b), 512));
    pool_att_bwd_kernel<<<grid, 256, 0, s>>>(*b, pool_att(p), A, lda, rinv, dA, acc);
    EAGCN_LAUNCH_CHECK();
    PoolAttGrad g;
    for (int k = 0; k < EAGCN_MAX_VIEWS; ++k) g.datt_w[k] = k < p->K ? p->datt_w[k] : nullptr;
    g.dave_a = p->dave_a;
    g.dself_r = p->dself_r;
    pool_att_bwd_final_kernel<<<1, 256, 0, s>>>(*b, pool_att(p), acc, g);
    EAGCN_LAUNCH_CHECK();
    return EAGCN_OK;
}

extern "C" int eagcn_pool_mix_forward(const eagcn_batch* b, const eagcn_layout* lay, const float* A, int lda,
                                      const float* padsum, const float* x, const float* pad_row, float* AX, int F,
                                      void* stream) {
    hipStream_t s = (hipStream_t)stream;
    EAGCN_CHECK_ARG(b && lay, "eagcn_pool_mix_forward: null argument");
    EAGCN_CHECK_ARG(b->T == 0 || (A && padsum && x && AX), "eagcn_pool_mix_forward: null buffer");
    EAGCN_CHECK_ARG(layout_width(lay) == F && F >= 1, "eagcn_pool_mix_forward: layout width %d != F %d", layout_width(lay), F);
    ProfScope ps(PROF_READOUT, s);
    pool_mix_fwd_kernel<<<dim3(b->B, cdiv(F, 64)), 256, 0, s>>>(*b, A, lda, padsum, x, make_colmap(lay), layout_ld(lay), pad_row,
                                                               AX, F);
    EAGCN_LAUNCH_CHECK();
    return EAGCN_OK;
}

extern "C" int eagcn_pool_mix_backward(const eagcn_batch* b, const eagcn_layout* lay, const float* A, int lda,
                                       const float* padsum, const float* x, const float* pad_row, const float* dAX, int F,
                                       float* dA, float* dx, float* dpad_row, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    EAGCN_CHECK_ARG(b && lay, "eagcn_pool_mix_backward: null argument");
    EAGCN_CHECK_ARG(b->T == 0 || (A && padsum && x && dAX), "eagcn_pool_mix_backward: null buffer");
    EAGCN_CHECK_ARG(layout_width(lay) == F && F >= 1, "eagcn_pool_mix_backward: layout width %d != F %d", layout_width(lay), F);
    ProfScope ps(PROF_READOUT, s);
    const ColMapD m = make_colmap(lay);
    const int ld = layout_ld(lay);
    if (dx) {
        pool_mix_bwd_x_kernel<<<dim3(b->B, cdiv(F, 64)), 256, 0, s>>>(*b, A, lda, dAX, m, ld, dx, F);
        EAGCN_LAUNCH_CHECK();
    }
    if (dA) {
        pool_mix_bwd_a_kernel<<<rows_grid(b), 256, 0, s>>>(*b, x, m, ld, pad_row, dAX, F, dA, lda);
        EAGCN_LAUNCH_CHECK();
    }
    if (dpad_row) {
        pool_mix_bwd_pad_kernel<<<cdiv(ld, 256), 256, 0, s>>>(*b, padsum, dAX, m, ld, F, dpad_row);
        EAGCN_LAUNCH_CHECK();
    }
    return EAGCN_OK;
}

extern "C" int eagcn_pool_reduce_forward(const eagcn_batch* b, const float* Z, int ldz, int F, int P, float* S, float* Pm,
                                         float* g, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    EAGCN_CHECK_ARG(b && Pm && g, "eagcn_pool_reduce_forward: null argument");
    EAGCN_CHECK_ARG(b->T == 0 || (Z && S), "eagcn_pool_reduce_forward: null buffer");
    EAGCN_CHECK_ARG(P >= 1 && P <= EAGCN_POOL_MAX, "eagcn_pool_reduce_forward: %d clusters (1..%d)", P, EAGCN_POOL_MAX);
    EAGCN_CHECK_ARG(F >= 1 && ldz >= F + P, "eagcn_pool_reduce_forward: ldz %d < F + P = %d", ldz, F + P);
    ProfScope ps(PROF_READOUT, s);
    pool_softmax_kernel<<<cdiv(std::max(b->T, 1), 256), 256, 0, s>>>(*b, Z, ldz, F, P, S);
    EAGCN_LAUNCH_CHECK();
    pool_reduce_fwd_kernel<<<dim3(b->B, cdiv(F, 64)), 256, 0, s>>>(*b, Z, ldz, F, P, S, Pm, g);
    EAGCN_LAUNCH_CHECK();
    return EAGCN_OK;
}

extern "C" int eagcn_pool_reduce_backward(const eagcn_batch* b, const float* Z, int ldz, int F, int P, const float* S,
                                          const float* Pm, const float* dg, float* dZ, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    EAGCN_CHECK_ARG(b && Pm && dg, "eagcn_pool_reduce_backward: null argument");
    EAGCN_CHECK_ARG(b->T == 0 || (Z && S && dZ), "eagcn_pool_reduce_backward: null buffer");
    EAGCN_CHECK_ARG(P >= 1 && P <= EAGCN_POOL_MAX, "eagcn_pool_reduce_backward: %d clusters (1..%d)", P, EAGCN_POOL_MAX);
    EAGCN_CHECK_ARG(F >= 1 && ldz >= F + P, "eagcn_pool_reduce_backward: ldz %d < F + P = %d", ldz, F + P);
    ProfScope ps(PROF_READOUT, s);
    pool_reduce_bwd_kernel<<<rows_grid(b), 256, 0, s>>>(*b, Z, ldz, F, P, S, Pm, dg, dZ);
    EAGCN_LAUNCH_CHECK();
    return EAGCN_OK;
}
