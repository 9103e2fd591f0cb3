// The GAT baseline layer of the reference (layers.py:99-203, used by models.py:69-73 for structure='GAT'): per molecule
//     h = X.W ;  e[i,j] = leakyrelu(a1.h_i + a2.h_j) ;  att = softmax_j( e[i,j] over j in adj_i + {i} )  (rows without a bond: 0)
//     att = dropout(att, 0.5) in training (layers.py:104,133) ;  h'_i = sum_j att[i,j] h_j ;  x = relu(dropout(h', p))  (:198-199)
// The reference builds the N x N x 2F pair tensor per molecule (layers.py:125); the attention is non-zero only at the bonds
// and the diagonal, so these kernels walk the bond lists of the batch index (eagcn_batch.row_ptr / col_ptr): scores per atom,
// a softmax over deg+1 entries per row, a gather of deg+1 rows of h.  A baseline layer: written for
// correctness and reasonable memory behaviour (16-lane groups per row, float4 columns), not tuned.
//
// Backward (softmax, leaky relu, both dropouts recomputed from the counter-based streams):
//     dq[i,j] = <dh'_i, h_j> ;  dp = dq * drop ;  de[i,j] = p[i,j] (dp[i,j] - sum_l p[i,l] dp[i,l]) ;  g = de * lrelu'
//     ds1_i = sum_j g[i,j] ;  ds2_j = sum_i g[i,j] ;  dh_j = sum_i q[i,j] dh'_i + ds1_j a1 + ds2_j a2
//     da1 = sum_i ds1_i h_i ;  da2 = sum_j ds2_j h_j ;  dW = X^T dh ;  dX = dh W^T
#include <algorithm>

#include <string.h>

#include "common.h"
#include "kernels.h"

namespace eagcn {

struct GatArgs {
    eagcn_batch bt;
    int F, Fp;
    const float* h; const float* a;
    float* s12;                  // [2][T]
    float alpha;
    int att_drop; uint32_t att_thr; float att_inv; uint64_t att_seed_;
    int do_drop; uint32_t thr; float inv_keep; uint64_t seed_;
    const uint64_t* seed_dev;    // the stream's seed lives in device memory (a captured launch draws fresh masks every replay)
    float* xout;
    const float* dxout;
    float* gbuf;                 // [E] per row-list entry: dp, then g
    float* gself;                // [T] the same for the diagonal entry
    float* stat;                 // [T][2] row max, row sum of exp
    float* ds;                   // [2][T]
    float* dh;                   // [T][Fp]
    float* da;                   // [2F]
};
constexpr uint64_t GAT_ATT_STREAM = 0xA77E17105EEDull;
__device__ __forceinline__ uint64_t gat_seed(const GatArgs& a) { return a.seed_dev ? *a.seed_dev : a.seed_; }
__device__ __forceinline__ uint64_t gat_att_seed(const GatArgs& a) { return a.seed_dev ? (*a.seed_dev ^ GAT_ATT_STREAM) : a.att_seed_; }


__device__ __forceinline__ float lrelu(float v, float alpha) { return v > 0.0f ? v : alpha * v; }
__device__ __forceinline__ float group_sum16(float v) {
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// s1 = h.a1, s2 = h.a2 per packed row (16 lanes per row)
__global__ __launch_bounds__(256) void gat_scores_kernel(GatArgs a) {
    const int T = dev_rows(a.bt);
    const int grp = threadIdx.x >> 4, sl = threadIdx.x & 15;
    for (int r = blockIdx.x * 16 + grp; r < T; r += gridDim.x * 16) {
        const float* hr = a.h + (size_t)r * a.Fp;
        const float v1 = group_sum16(dot16(hr, a.a, sl, a.F));
        const float v2 = group_sum16(dot16(hr, a.a + a.F, sl, a.F));
        if (sl == 0) { a.s12[r] = v1; a.s12[a.bt.T + r] = v2; }
    }
}

// softmax statistics of row r over its bonds and the diagonal (all lanes compute the same scalars)
struct RowSoft { float mx, Z; };
__device__ __forceinline__ RowSoft gat_row_soft(const GatArgs& a, int r, int r0, int iloc, int2 rp, float s1i) {
    const float* s2 = a.s12 + a.bt.T;
    float mx = lrelu(s1i + s2[r], a.alpha);
    for (int e = 0; e < rp.y; ++e) {
        const int jl = a.bt.nbr[rp.x + e];
        if (jl != iloc) mx = fmaxf(mx, lrelu(s1i + s2[r0 + jl], a.alpha));
    }
    float Z = __expf(lrelu(s1i + s2[r], a.alpha) - mx);
    for (int e = 0; e < rp.y; ++e) {
        const int jl = a.bt.nbr[rp.x + e];
        if (jl != iloc) Z += __expf(lrelu(s1i + s2[r0 + jl], a.alpha) - mx);
    }
    return RowSoft{mx, Z};
}
// attention weight of entry (row r -> local column jl) after the attention dropout
__device__ __forceinline__ float gat_q(const GatArgs& a, int r, int jl, float s1i, float s2j, const RowSoft& rs) {
    const float p = __expf(lrelu(s1i + s2j, a.alpha) - rs.mx) / rs.Z;
    return a.att_drop ? p * drop_scale(gat_att_seed(a), (uint64_t)r * 1024ull + (uint64_t)jl, a.att_thr, a.att_inv) : p;
}

__global__ __launch_bounds__(256) void gat_attend_fwd_kernel(GatArgs a) {
    const eagcn_batch& bt = a.bt;
    const int T = dev_rows(bt);
    const int grp = threadIdx.x >> 4, sl = threadIdx.x & 15;
    const float* s2 = a.s12 + bt.T;
    for (int r = blockIdx.x * 16 + grp; r < T; r += gridDim.x * 16) {
        const int4 info = reinterpret_cast<const int4*>(bt.row_info)[r];
        const int iloc = info.y, r0 = info.w;
        const int2 rp = reinterpret_cast<const int2*>(bt.row_ptr)[r];
        float* xo = a.xout + (size_t)r * a.Fp;
        if (bt.row_m[r] == 0.0f) {                                  // no bond: the reference's attention row is all zero
            for (int c = sl * 4; c < a.Fp; c += 64) *reinterpret_cast<float4*>(xo + c) = make_float4(0.f, 0.f, 0.f, 0.f);
            continue;
        }
        const float s1i = a.s12[r];
        const RowSoft rs = gat_row_soft(a, r, r0, iloc, rp, s1i);
        const float qself = gat_q(a, r, iloc, s1i, s2[r], rs);
        for (int c = sl * 4; c < a.Fp; c += 64) {
            const float4 hs = *reinterpret_cast<const float4*>(a.h + (size_t)r * a.Fp + c);
            float4 acc = make_float4(qself * hs.x, qself * hs.y, qself * hs.z, qself * hs.w);
            for (int e = 0; e < rp.y; ++e) {
                const int jl = bt.nbr[rp.x + e];
                if (jl == iloc) continue;
                const float q = gat_q(a, r, jl, s1i, s2[r0 + jl], rs);
                const float4 hj = *reinterpret_cast<const float4*>(a.h + (size_t)(r0 + jl) * a.Fp + c);
                acc.x = fmaf(q, hj.x, acc.x); acc.y = fmaf(q, hj.y, acc.y); acc.z = fmaf(q, hj.z, acc.z); acc.w = fmaf(q, hj.w, acc.w);
            }
            float o[4] = {fmaxf(acc.x, 0.0f), fmaxf(acc.y, 0.0f), fmaxf(acc.z, 0.0f), fmaxf(acc.w, 0.0f)};
            if (a.do_drop) {
#pragma unroll
                for (int u = 0; u < 4; ++u) o[u] *= drop_scale(gat_seed(a), (uint64_t)r * a.Fp + c + u, a.thr, a.inv_keep);
            }
            *reinterpret_cast<float4*>(xo + c) = make_float4(o[0], o[1], o[2], o[3]);
        }
    }
}

// dh'_r[c] = dxout * dropout scale * [relu active]   (xout > 0 <=> relu active and kept)
__device__ __forceinline__ float gat_dhp(const GatArgs& a, int r, int c) {
    const size_t o = (size_t)r * a.Fp + c;
    if (!(a.xout[o] > 0.0f)) return 0.0f;
    const float ds = a.do_drop ? drop_scale(gat_seed(a), (uint64_t)r * a.Fp + c, a.thr, a.inv_keep) : 1.0f;
    return a.dxout[o] * ds;
}

__global__ __launch_bounds__(256) void gat_bwd_rows_kernel(GatArgs a) {
    const eagcn_batch& bt = a.bt;
    const int T = dev_rows(bt);
    const int grp = threadIdx.x >> 4, sl = threadIdx.x & 15;
    const int lane0 = (threadIdx.x & 63) & ~15;                      // first lane of this 16-lane group inside its wave
    const float* s2 = a.s12 + bt.T;
    for (int r = blockIdx.x * 16 + grp; r < T; r += gridDim.x * 16) {
        const int4 info = reinterpret_cast<const int4*>(bt.row_info)[r];
        const int iloc = info.y, r0 = info.w;
        const int2 rp = reinterpret_cast<const int2*>(bt.row_ptr)[r];
        if (bt.row_m[r] == 0.0f) {
            if (sl == 0) { a.ds[r] = 0.0f; a.gself[r] = 0.0f; a.stat[2 * r] = 0.0f; a.stat[2 * r + 1] = 1.0f; }
            continue;
        }
        const float s1i = a.s12[r];
        const RowSoft rs = gat_row_soft(a, r, r0, iloc, rp, s1i);
        if (sl == 0) { a.stat[2 * r] = rs.mx; a.stat[2 * r + 1] = rs.Z; }
        // pass A: dp of every entry (the diagonal first), row dot
        float rowdot = 0.0f, dp_self = 0.0f;
        for (int e = -1; e < rp.y; ++e) {
            const int jl = e < 0 ? iloc : bt.nbr[rp.x + e];
            if (e >= 0 && jl == iloc) { if (sl == 0) a.gbuf[rp.x + e] = 0.0f; continue; }
            const float* hj = a.h + (size_t)(r0 + jl) * a.Fp;
            float part = 0.0f;
            for (int c = sl; c < a.F; c += 16) part += gat_dhp(a, r, c) * hj[c];
            const float dq = group_sum16(part);
            const float pre = s1i + s2[r0 + jl];
            const float p = __expf(lrelu(pre, a.alpha) - rs.mx) / rs.Z;
            const float d = a.att_drop ? drop_scale(gat_att_seed(a), (uint64_t)r * 1024ull + (uint64_t)jl, a.att_thr, a.att_inv) : 1.0f;
            const float dp = dq * d;
            rowdot += p * dp;
            if (e < 0) dp_self = dp;
            else if (sl == 0) a.gbuf[rp.x + e] = dp;
        }
        // pass B: g = p (dp - rowdot) * lrelu'
        float ds1 = 0.0f;
        for (int e = -1; e < rp.y; ++e) {
            const int jl = e < 0 ? iloc : bt.nbr[rp.x + e];
            if (e >= 0 && jl == iloc) continue;
            float dp = dp_self;
            if (e >= 0) {
                dp = sl == 0 ? a.gbuf[rp.x + e] : 0.0f;              // lane 0 reads back what it wrote
                dp = __shfl(dp, lane0);
            }
            const float pre = s1i + s2[r0 + jl];
            const float p = __expf(lrelu(pre, a.alpha) - rs.mx) / rs.Z;
            const float g = p * (dp - rowdot) * (pre > 0.0f ? 1.0f : a.alpha);
            ds1 += g;
            if (sl == 0) { if (e < 0) a.gself[r] = g; else a.gbuf[rp.x + e] = g; }
        }
        if (sl == 0) a.ds[r] = ds1;
    }
}

__global__ __launch_bounds__(256) void gat_bwd_cols_kernel(GatArgs a) {
    const eagcn_batch& bt = a.bt;
    const int T = dev_rows(bt);
    const int grp = threadIdx.x >> 4, sl = threadIdx.x & 15;
    const float* s2 = a.s12 + bt.T;
    for (int r = blockIdx.x * 16 + grp; r < T; r += gridDim.x * 16) {
        const int4 info = reinterpret_cast<const int4*>(bt.row_info)[r];
        const int jloc = info.y, r0 = info.w;
        const int2 cp = reinterpret_cast<const int2*>(bt.col_ptr)[r];
        // ds2_j = g of the diagonal + g of every bond into j (looked up in the source row's list)
        float ds2 = a.gself[r];
        for (int t = 0; t < cp.y; ++t) {
            const int il = bt.tnbr[cp.x + t];
            if (il == jloc) continue;
            const int2 rpi = reinterpret_cast<const int2*>(bt.row_ptr)[r0 + il];
            for (int e = 0; e < rpi.y; ++e)
                if (bt.nbr[rpi.x + e] == jloc) { ds2 += a.gbuf[rpi.x + e]; break; }
        }
        if (sl == 0) a.ds[bt.T + r] = ds2;
        const float ds1 = a.ds[r];
        const bool live = bt.row_m[r] != 0.0f;
        float qself = 0.0f;
        if (live) {
            const RowSoft rs{a.stat[2 * r], a.stat[2 * r + 1]};
            qself = gat_q(a, r, jloc, a.s12[r], s2[r], rs);
        }
        float* out = a.dh + (size_t)r * a.Fp;
        for (int c = sl; c < a.Fp; c += 16) {
            float acc = 0.0f;
            if (c < a.F) {
                acc = ds1 * a.a[c] + ds2 * a.a[a.F + c];
                if (live) acc += qself * gat_dhp(a, r, c);
                for (int t = 0; t < cp.y; ++t) {
                    const int il = bt.tnbr[cp.x + t];
                    if (il == jloc) continue;
                    const int ri = r0 + il;
                    const RowSoft rsi{a.stat[2 * ri], a.stat[2 * ri + 1]};
                    acc += gat_q(a, ri, jloc, a.s12[ri], s2[r], rsi) * gat_dhp(a, ri, c);
                }
            }
            out[c] = acc;
        }
    }
}

// da1[c] = sum_r ds1[r] h[r][c], da2[c] = sum_r ds2[r] h[r][c]: a workgroup owns 16 columns, 16 row lanes, fixed-order sums
__global__ __launch_bounds__(256) void gat_da_kernel(GatArgs a) {
    __shared__ double red[2][16][16];
    const int T = dev_rows(a.bt);
    const int cl = threadIdx.x & 15, rl = threadIdx.x >> 4;
    const int c = blockIdx.x * 16 + cl;
    double t1 = 0.0, t2 = 0.0;
    if (c < a.F)
        for (int r = rl; r < T; r += 16) {
            const float hv = a.h[(size_t)r * a.Fp + c];
            t1 += (double)(a.ds[r] * hv);
            t2 += (double)(a.ds[a.bt.T + r] * hv);
        }
    red[0][rl][cl] = t1;
    red[1][rl][cl] = t2;
    __syncthreads();
    if (rl < 2 && c < a.F) {
        double t = 0.0;
        for (int q = 0; q < 16; ++q) t += red[rl][q][cl];
        a.da[rl * a.F + c] = (float)t;
    }
}

struct GatScratch { float *gbuf, *gself, *stat, *ds, *dh; };
static size_t gat_carve(void* base, const eagcn_batch* b, int Fp, GatScratch* out) {
    char* p = (char*)base;
    size_t off = 0;
    auto take = [&](size_t n) { float* q = base ? (float*)(p + off) : nullptr; off = align256(off + std::max<size_t>(n, 1) * sizeof(float)); return q; };
    GatScratch s;
    const size_t T = (size_t)std::max(b->T, 1);
    s.gbuf = take((size_t)std::max(b->E, 1));
    s.gself = take(T);
    s.stat = take(2 * T);
    s.ds = take(2 * T);
    s.dh = take(T * Fp);
    if (out) *out = s;
    return off;
}

static int gat_check(const eagcn_batch* b, const eagcn_gat_params* p, const char* who) {
    EAGCN_CHECK_ARG(b && p, "%s: null argument", who);
    EAGCN_CHECK_ARG(p->fin >= 1 && p->ld_in >= p->fin && p->F >= 1 && p->W && p->a, "%s: bad layer description", who);
    EAGCN_CHECK_ARG(b->row_ptr && b->col_ptr && b->nbr && b->tnbr && b->build_lists,
                    "%s: the batch index was built without bond lists (eagcn_batch.build_lists)", who);
    EAGCN_CHECK_ARG(b->N <= 1024, "%s: N=%d exceeds the supported 1024 atoms", who, b->N);
    EAGCN_CHECK_ARG(p->dropout >= 0.0f && p->dropout < 1.0f && p->att_dropout >= 0.0f && p->att_dropout < 1.0f, "%s: dropout out of [0,1)", who);
    return EAGCN_OK;
}

static GatArgs gat_args(const eagcn_batch* b, const eagcn_gat_params* p, const float* h, float* s12) {
    GatArgs a;
    memset(&a, 0, sizeof(a));
    a.bt = *b; a.F = p->F; a.Fp = pad16(p->F); a.h = h; a.a = p->a; a.s12 = s12; a.alpha = p->alpha;
    a.att_drop = (p->training && p->att_dropout > 0.0f) ? 1 : 0;
    a.att_thr = (uint32_t)std::min(4294967295.0, (double)p->att_dropout * 4294967296.0);
    a.att_inv = 1.0f / (1.0f - p->att_dropout);
    a.att_seed_ = p->seed ^ GAT_ATT_STREAM;
    a.do_drop = (p->training && p->dropout > 0.0f) ? 1 : 0;
    a.thr = (uint32_t)std::min(4294967295.0, (double)p->dropout * 4294967296.0);
    a.inv_keep = 1.0f / (1.0f - p->dropout);
    a.seed_ = p->seed;
    a.seed_dev = p->seed_dev;
    return a;
}
static inline int row_grid(int T) { return std::max(1, std::min(cdiv(std::max(T, 1), 16), 2048)); }

}  // namespace eagcn

using namespace eagcn;

extern "C" size_t eagcn_gat_scratch_bytes(const eagcn_batch* b, int F) { return b ? gat_carve(nullptr, b, pad16(F), nullptr) : 0; }

extern "C" int eagcn_gat_forward(const eagcn_batch* b, const eagcn_gat_params* p, const float* x, float* h, float* s12,
                                 float* xout, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    int rc = gat_check(b, p, "eagcn_gat_forward");
    if (rc) return rc;
    if (b->T == 0) return EAGCN_OK;
    EAGCN_CHECK_ARG(x && h && s12 && xout, "eagcn_gat_forward: null buffer");
    const int Fp = pad16(p->F);
    rc = zero_fill(h, (size_t)b->T * Fp * sizeof(float), s);                       // (padding columns of h stay zero)
    if (rc) return rc;
    GemmDesc g{0, 0, b->T, p->F, p->fin, x, p->ld_in, p->W, p->F, h, Fp, 1, 0};
    g.M_dev = b->meta + EAGCN_META_T;
    rc = launch_gemm(g, s);
    if (rc) return rc;
    GatArgs a = gat_args(b, p, h, s12);
    a.xout = xout;
    ProfScope ps(PROF_AGG, s);
    gat_scores_kernel<<<row_grid(b->T), 256, 0, s>>>(a);
    EAGCN_LAUNCH_CHECK();
    gat_attend_fwd_kernel<<<row_grid(b->T), 256, 0, s>>>(a);
    EAGCN_LAUNCH_CHECK();
    return EAGCN_OK;
}

extern "C" int eagcn_gat_backward(const eagcn_batch* b, const eagcn_gat_params* p, const float* x, const float* h,
                                  const float* s12, const float* xout, const float* dxout, float* dx, float* dW, float* da,
                                  void* scratch, size_t scratch_bytes, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    int rc = gat_check(b, p, "eagcn_gat_backward");
    if (rc) return rc;
    EAGCN_CHECK_ARG(dW && da, "eagcn_gat_backward: null gradient buffer");
    if (b->T == 0) {
        rc = zero_fill(dW, (size_t)p->fin * p->F * sizeof(float), s);
        if (rc) return rc;
        rc = zero_fill(da, (size_t)2 * p->F * sizeof(float), s);
        if (rc) return rc;
        return EAGCN_OK;
    }
    EAGCN_CHECK_ARG(x && h && s12 && xout && dxout && scratch, "eagcn_gat_backward: null buffer");
    const int Fp = pad16(p->F);
    GatScratch sc;
    EAGCN_CHECK_ARG(gat_carve(scratch, b, Fp, &sc) <= scratch_bytes, "eagcn_gat_backward: scratch too small");
    GatArgs a = gat_args(b, p, h, const_cast<float*>(s12));
    a.xout = const_cast<float*>(xout); a.dxout = dxout;
    a.gbuf = sc.gbuf; a.gself = sc.gself; a.stat = sc.stat; a.ds = sc.ds; a.dh = sc.dh; a.da = da;
    {
        ProfScope ps(PROF_EDGE, s);
        gat_bwd_rows_kernel<<<row_grid(b->T), 256, 0, s>>>(a);
        EAGCN_LAUNCH_CHECK();
        gat_bwd_cols_kernel<<<row_grid(b->T), 256, 0, s>>>(a);
        EAGCN_LAUNCH_CHECK();
        gat_da_kernel<<<cdiv(p->F, 16), 256, 0, s>>>(a);
        EAGCN_LAUNCH_CHECK();
    }
    GemmDesc gw{1, 0, p->fin, p->F, b->T, x, p->ld_in, sc.dh, Fp, dW, p->F, 1, 0};
    gw.K_dev = b->meta + EAGCN_META_T;
    rc = launch_gemm(gw, s);
    if (rc) return rc;
    if (dx) {
        rc = zero_fill(dx, (size_t)b->T * p->ld_in * sizeof(float), s);               // padding columns of dx
        if (rc) return rc;
        GemmDesc gx{0, 1, b->T, p->fin, p->F, sc.dh, Fp, p->W, p->F, dx, p->ld_in, 1, 0};
        gx.M_dev = b->meta + EAGCN_META_T;
        rc = launch_gemm(gx, s);
        if (rc) return rc;
    }
    return EAGCN_OK;
}
