// One multi-view graph-conv layer, forward and backward: parameter packing, the per-view
// BatchNorm over all B*N_pad rows (reference layers.py:408-412) with its grid-wide reduction, the
// relu / dropout / mask / view-merge epilogue (layers.py:93-94, 313-316), and the orchestration of
// the GEMM and aggregation kernels behind the C ABI.
//
// BatchNorm over rows that are not stored: the reference normalises over every row of the padded
// [B,N,F] tensor.  A padded (or bond-less) row has an all-zero attention row, so its pre-BN value
// is exactly the bias.  Working with y' = y - bias (the GEMM output without bias) those rows are
// exactly 0: they add nothing to sum(y') and sum(y'^2) and only enter through the row count
// M = B*N.  The bias cancels inside a training-mode BatchNorm; it re-enters in the running mean
// and in eval mode.
#include <algorithm>

#include <stdlib.h>

#include "common.h"
#include "kernels.h"

namespace eagcn {

struct ParamPtrs {
    const float* att_w[EAGCN_MAX_VIEWS];
    const float* self_r[EAGCN_MAX_VIEWS];
    const float* W[EAGCN_MAX_VIEWS];
    const float* bias[EAGCN_MAX_VIEWS];
    const float* gamma[EAGCN_MAX_VIEWS];
    const float* beta[EAGCN_MAX_VIEWS];
    float* run_mean[EAGCN_MAX_VIEWS];
    float* run_var[EAGCN_MAX_VIEWS];
    const float* ave_w;
    int channels[EAGCN_MAX_VIEWS];          // bond-type codes of view k (1..channels[k])
    const float* rel_vec[EAGCN_MAX_VIEWS];  // general relations: code c+1 -> channel vector rel_vec[k][c][0..rel_c[k]); null: one-hot
    int rel_c[EAGCN_MAX_VIEWS];
};
// attention logit of bond code c (1-based) in view k: w[c-1] for one-hot relation tensors, <w, vec[c-1]> in general (layers.py:82)
__device__ __forceinline__ float att_logit(const ParamPtrs& pp, int k, int c) {
    if (!pp.rel_vec[k]) return pp.att_w[k][c - 1];
    const float* v = pp.rel_vec[k] + (size_t)(c - 1) * pp.rel_c[k];
    float s = 0.0f;
    for (int ch = 0; ch < pp.rel_c[k]; ++ch) s = fmaf(pp.att_w[k][ch], v[ch], s);
    return s;
}
struct GradPtrs {
    float* dW[EAGCN_MAX_VIEWS];
    float* dbias[EAGCN_MAX_VIEWS];
    float* dgamma[EAGCN_MAX_VIEWS];
    float* dbeta[EAGCN_MAX_VIEWS];
    float* datt_w[EAGCN_MAX_VIEWS];
    float* dself_r[EAGCN_MAX_VIEWS];
    float* dave_w;
};

// colp rows
enum { CP_GAMMA = 0, CP_BETA, CP_BIAS, CP_RMEAN, CP_RVAR, CP_AVEW, CP_ROWS = 8 };
// bn rows

__device__ __forceinline__ int col_view(const ViewCols& vc, int cp) {
    int k = 0;
#pragma unroll
    for (int v = 1; v < EAGCN_MAX_VIEWS; ++v) k += (v < vc.K && cp >= vc.off[v]) ? 1 : 0;
    return k;
}
__device__ __forceinline__ int packed_to_exact(const ColMapD& m, int cp) {
    int eo = 0, po = 0;
    for (int s = 0; s < m.nseg; ++s) {
        if (cp < po + m.p[s]) return (cp - po < m.w[s]) ? eo + (cp - po) : -1;
        eo += m.w[s];
        po += m.p[s];
    }
    return -1;
}

// ---- parameter packing -------------------------------------------------------------------------
__device__ __forceinline__ void pack_params_body(const ParamPtrs& pp, const ViewCols& vc, const ColMapD& in, int ld_in,
                                                 int fp, float* __restrict__ Wcat, float* __restrict__ WcatT,
                                                 float* __restrict__ colp, float* __restrict__ sig, float* __restrict__ rsig,
                                                 const BxOut wp = BxOut{nullptr, 0, 0, 0}, const BxOut wtp = BxOut{nullptr, 0, 0, 0}) {
    // 32 x 32 tiles through LDS: Wcat rows AND the rows of its transpose are written as contiguous 128-byte runs (the
    // transpose used to leave as 4-byte stores ld_in floats apart: one cache line per lane)
    __shared__ float tile[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;              // 32 columns x 8 rows per pass
    const int ntx = (fp + 31) >> 5, nty = (ld_in + 31) >> 5;
    for (int t = blockIdx.x; t < ntx * nty; t += gridDim.x) {
        const int ip0 = (t / ntx) << 5, cp0 = (t % ntx) << 5;
        {
            const int cp = cp0 + tx;
            const int k = cp < fp ? col_view(vc, cp) : 0;
            const int f = cp - vc.off[k];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int ip = ip0 + ty + 8 * j;
                float v = 0.0f;
                if (ip < ld_in && cp < fp) {
                    const int fi = packed_to_exact(in, ip);
                    if (fi >= 0 && f < vc.width[k]) v = pp.W[k][(size_t)fi * vc.width[k] + f];
                    Wcat[(size_t)ip * fp + cp] = v;
                }
                tile[ty + 8 * j][tx] = v;
            }
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int cp = cp0 + ty + 8 * j, ip = ip0 + tx;
            // [Fp][ld_in]: the K-contiguous B operand of the forward product (NT form)
            if (cp < fp && ip < ld_in) WcatT[(size_t)cp * ld_in + ip] = tile[tx][ty + 8 * j];
        }
        // operand planes of the plane GEMMs (gemm_bx3.hip) of both matrices, four adjacent elements per thread (8-byte stores:
        // scalar 2-byte stores made this launch four times as long); fp and ld_in are multiples of 16 for every layer that has planes
        if (wp.p) {
            const int r = threadIdx.x >> 3, c4 = (threadIdx.x & 7) << 2;
            if (ip0 + r < ld_in && cp0 + c4 < fp)
                bx_store4(wp, ip0 + r, cp0 + c4, make_float4(tile[r][c4], tile[r][c4 + 1], tile[r][c4 + 2], tile[r][c4 + 3]));
            if (cp0 + r < fp && ip0 + c4 < ld_in)
                bx_store4(wtp, cp0 + r, ip0 + c4, make_float4(tile[c4][r], tile[c4 + 1][r], tile[c4 + 2][r], tile[c4 + 3][r]));
        }
        __syncthreads();
    }
    for (int cp = blockIdx.x * blockDim.x + threadIdx.x; cp < fp; cp += gridDim.x * blockDim.x) {
        const int k = col_view(vc, cp), f = cp - vc.off[k];
        const bool ok = f < vc.width[k];
        colp[CP_GAMMA * fp + cp] = ok ? pp.gamma[k][f] : 0.0f;
        colp[CP_BETA * fp + cp] = ok ? pp.beta[k][f] : 0.0f;
        colp[CP_BIAS * fp + cp] = ok ? pp.bias[k][f] : 0.0f;
        colp[CP_RMEAN * fp + cp] = ok ? pp.run_mean[k][f] : 0.0f;
        colp[CP_RVAR * fp + cp] = ok ? pp.run_var[k][f] : 1.0f;
        colp[CP_AVEW * fp + cp] = pp.ave_w ? pp.ave_w[k] : 1.0f;
    }
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < vc.K * 256; e += gridDim.x * blockDim.x) {
        const int k = e >> 8, c = e & 255;
        sig[e] = (c >= 1 && c <= pp.channels[k]) ? sigmoidf_(att_logit(pp, k, c)) : 0.0f;
    }
    if (blockIdx.x == 0 && threadIdx.x < vc.K) rsig[threadIdx.x] = sigmoidf_(pp.self_r[threadIdx.x][0]);
}
__global__ __launch_bounds__(256) void pack_params_kernel(ParamPtrs pp, ViewCols vc, ColMapD in, int ld_in, int fp,
                                                           float* __restrict__ Wcat, float* __restrict__ WcatT,
                                                           float* __restrict__ colp, float* __restrict__ sig,
                                                           float* __restrict__ rsig, BxOut wp, BxOut wtp) {
    pack_params_body(pp, vc, in, ld_in, fp, Wcat, WcatT, colp, sig, rsig, wp, wtp);
}
// every layer of a model in one launch (blockIdx.y = layer): the parameters of all layers are known before the
// first layer runs, and each packing launch is a few microseconds of fixed cost on the critical path
struct PackJob {
    ParamPtrs pp; ViewCols vc; ColMapD in; int ld_in, fp;
    float *Wcat, *WcatT, *colp, *sig, *rsig;
    BxOut wp, wtp;
};
struct PackJobs { PackJob j0, j1, j2, j3; };
__global__ __launch_bounds__(256) void pack_params_multi_kernel(PackJobs jobs, ZeroJob zj, HandOff ho) {
    // the call prologue of the model engine rides along: hand-off flags of gemm3.hip and the head's fp64 BatchNorm sums
    if (blockIdx.y == 0) {
        for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < max(zj.nu, zj.nd); i += gridDim.x * blockDim.x) {
            if (i < zj.nu) zj.u[i] = 0u;
            if (i < zj.nd) zj.d[i] = 0.0;
        }
    }
    // (an if-chain, not an indexed array: indexing the by-value argument block dynamically would move it to scratch)
    if (blockIdx.y == 0) pack_params_body(jobs.j0.pp, jobs.j0.vc, jobs.j0.in, jobs.j0.ld_in, jobs.j0.fp, jobs.j0.Wcat, jobs.j0.WcatT, jobs.j0.colp, jobs.j0.sig, jobs.j0.rsig, jobs.j0.wp, jobs.j0.wtp);
    else if (blockIdx.y == 1) pack_params_body(jobs.j1.pp, jobs.j1.vc, jobs.j1.in, jobs.j1.ld_in, jobs.j1.fp, jobs.j1.Wcat, jobs.j1.WcatT, jobs.j1.colp, jobs.j1.sig, jobs.j1.rsig, jobs.j1.wp, jobs.j1.wtp);
    else if (blockIdx.y == 2) pack_params_body(jobs.j2.pp, jobs.j2.vc, jobs.j2.in, jobs.j2.ld_in, jobs.j2.fp, jobs.j2.Wcat, jobs.j2.WcatT, jobs.j2.colp, jobs.j2.sig, jobs.j2.rsig, jobs.j2.wp, jobs.j2.wtp);
    else pack_params_body(jobs.j3.pp, jobs.j3.vc, jobs.j3.in, jobs.j3.ld_in, jobs.j3.fp, jobs.j3.Wcat, jobs.j3.WcatT, jobs.j3.colp, jobs.j3.sig, jobs.j3.rsig, jobs.j3.wp, jobs.j3.wtp);
    // ... and the step's stream hand-offs (kernels.h HandOff): this launch reads parameters only, the batch is needed from the next one on
    if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) handoff_body(ho);
}

// sum of per-workgroup partial pairs slab[s][cp][0..1] over s, L (16 or 64) lanes per column: with hundreds of
// slabs, 64 lanes per column turn five dependent batches of loads per lane into one or two
template <int L>
__device__ __forceinline__ void slab_sum(const double* __restrict__ slab, int nslab, int fp, int cp, int sl,
                                           double& s1, double& s2) {
    s1 = 0.0;
    s2 = 0.0;
    for (int s0 = sl; s0 < nslab; s0 += L * 8) {           // 8 slab pairs in flight per lane
        double2 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int s = s0 + L * u;
            v[u] = s < nslab ? *reinterpret_cast<const double2*>(slab + ((size_t)s * fp + cp) * 2) : make_double2(0.0, 0.0);
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) { s1 += v[u].x; s2 += v[u].y; }
    }
#pragma unroll
    for (int o = L / 2; o > 0; o >>= 1) {
        s1 += __shfl_xor(s1, o);
        s2 += __shfl_xor(s2, o);
    }
}

// ---- sync-BatchNorm (SURVEY.md 8e "BN modes (ii)"): this rank's per-channel sums as ONE vector out[2 Fp + 1] = (sum, sum of
//      squares / sum dH, sum dH xhat per column; rows of the BatchNorm) that the caller's hook all-reduces across the ranks; the
//      finalize kernels then read the reduced vector (forward: as a single slab with the global row count; backward: for the
//      means c1, c2 only -- d gamma / d beta stay this rank's sums, the gradient all-reduce averages them like every gradient).
//      mode 1: forward slabs (live count from the tile count), mode 2: backward slabs (from the row count)
constexpr int BWD_ROWS = 7;
template <int L>
__global__ __launch_bounds__(256) void bn_stats_sum_kernel(const double* __restrict__ slab, int nslab, int fp, double M,
                                                            double* __restrict__ out, const int32_t* __restrict__ meta,
                                                            int mode, int tiles_per_wg, int nvirt, int batch_B) {
    if (meta[EAGCN_META_NLOG] > 0) M = (double)batch_B * (double)meta[EAGCN_META_NLOG];
    if (mode == 1 && tiles_per_wg > 0) nslab = min(nslab, (meta[EAGCN_META_NTILES] + tiles_per_wg - 1) / tiles_per_wg);
    if (mode == 1 && tiles_per_wg < 0) nslab = min(nslab, meta[EAGCN_META_NBLK]);      // (lagg.hip: one slab per row block)
    if (mode == 2) nslab = max(1, min(nslab, (meta[EAGCN_META_T] + nvirt + BWD_ROWS - 1) / BWD_ROWS));
    const int cpr = blockIdx.x * (256 / L) + threadIdx.x / L, sl = threadIdx.x % L;
    const int cp = min(cpr, fp - 1);
    double s1, s2;
    slab_sum<L>(slab, nslab, fp, cp, sl, s1, s2);
    if (cpr < fp && sl == 0) { out[2 * cp] = s1; out[2 * cp + 1] = s2; }
    if (blockIdx.x == 0 && threadIdx.x == 0) out[2 * fp] = M;
}

// ---- BatchNorm forward ---------------------------------------------------------------------------
// 256/L columns per workgroup, L lanes per column
template <int L>
__global__ __launch_bounds__(256) void bn_finalize_kernel(const double* __restrict__ slab, int nslab, int fp,
                                                           double M, int training, float eps, float momentum,
                                                           const float* __restrict__ colp, ParamPtrs pp,
                                                           ViewCols vc, float* __restrict__ bn,
                                                           const int32_t* __restrict__ meta, int tiles_per_wg, int batch_B,
                                                           const double* __restrict__ M_dev) {
    // aggregation workgroups beyond the actual tile count exit without writing their slab
    if (meta[EAGCN_META_NLOG] > 0) M = (double)batch_B * (double)meta[EAGCN_META_NLOG];    // (M was computed from the capacity N)
    if (M_dev) M = *M_dev;                             // sync-BatchNorm: rows of ALL ranks (slab = the all-reduced sums)
    if (tiles_per_wg > 0) nslab = min(nslab, (meta[EAGCN_META_NTILES] + tiles_per_wg - 1) / tiles_per_wg);   // (0: every slab is written)
    if (tiles_per_wg < 0) nslab = min(nslab, meta[EAGCN_META_NBLK]);                    // (lagg.hip: one slab per row block)
    const int cpr = blockIdx.x * (256 / L) + threadIdx.x / L, sl = threadIdx.x % L;
    const int cp = min(cpr, fp - 1);
    double s1 = 0.0, s2 = 0.0;
    if (training) slab_sum<L>(slab, nslab, fp, cp, sl, s1, s2);
    if (cpr >= fp || sl != 0) return;
    const int k = col_view(vc, cp), f = cp - vc.off[k];
    const float gamma = colp[CP_GAMMA * fp + cp], beta = colp[CP_BETA * fp + cp];
    const float bias = colp[CP_BIAS * fp + cp];
    float mu, inv;
    if (training) {
        const double mean = s1 / M;
        double var = s2 / M - mean * mean;
        var = var > 0.0 ? var : 0.0;
        mu = (float)mean;
        inv = (float)(1.0 / sqrt(var + (double)eps));
        if (f < vc.width[k]) {
            const double unbiased = var * (M / (M - 1.0));
            pp.run_mean[k][f] = (float)((1.0 - momentum) * (double)colp[CP_RMEAN * fp + cp] + momentum * (mean + (double)bias));
            pp.run_var[k][f] = (float)((1.0 - momentum) * (double)colp[CP_RVAR * fp + cp] + momentum * unbiased);
        }
    } else {
        mu = colp[CP_RMEAN * fp + cp] - bias;
        inv = 1.0f / sqrtf(colp[CP_RVAR * fp + cp] + eps);
    }
    const float sc = gamma * inv;
    bn[BN_SC * fp + cp] = sc;
    bn[BN_SH * fp + cp] = beta - mu * sc;
    bn[BN_MU * fp + cp] = mu;
    bn[BN_INV * fp + cp] = inv;
}

struct ApplyArgs {
    eagcn_batch bt;
    ViewCols vc;
    int structure, fp;
    const float* Y; int ldy;
    const float* bn;
    const float* colp;
    float* out; int ldo;
    float* pad_row;
    int do_drop; uint32_t thr; float inv_keep; uint64_t seed; const uint64_t* seed_dev;
    BxOut planes;                // optional: the operand planes of the next layer's products (gemm_bx3.hip), written here
    int tile_map;                // element -> lane map: 0: row-major groups of four columns (a wavefront = 256 columns of one row: eight
                                 // 64-byte pieces per plane image); 1: a wavefront = 8 rows x one 32-column panel = 512 CONTIGUOUS bytes
                                 // of every plane image (bx3.h: panel-major) and eight 128-byte row segments of the fp32 matrices
};

__global__ __launch_bounds__(256) void bn_apply_kernel(ApplyArgs a) {
    const int fp = a.fp;
    const uint64_t seed = a.seed_dev ? *a.seed_dev : a.seed;   // (never write to the by-value argument block:
                                                               //  that would demote it to scratch memory)
    if (blockIdx.x == 0) {   // value of the rows that are not stored
        for (int c = threadIdx.x; c < a.ldo; c += blockDim.x) {
            float v = 0.0f;
            if (a.structure == EAGCN_STRUCT_WEIGHTED)
                for (int k = 0; k < a.vc.K; ++k) {
                    const int cp = a.vc.off[k] + c;
                    v += a.colp[CP_AVEW * fp + cp] * fmaxf(a.bn[BN_SH * fp + cp], 0.0f);
                }
            a.pad_row[c] = v;
        }
    }
    const int T = dev_rows(a.bt);
    // (element index -> (row, column group) with a 32-bit division: T * columns / 4 of any batch that fits the index fits 31 bits)
    if (a.structure == EAGCN_STRUCT_CONCATE) {
        const uint32_t g4 = a.tile_map ? ((uint32_t)(fp + 31) >> 5) : fp / 4;
        const uint32_t total = a.tile_map ? (((uint32_t)T + 7) >> 3) * g4 * 64u : (uint32_t)T * g4;
        for (uint32_t e = blockIdx.x * blockDim.x + threadIdx.x; e < total; e += gridDim.x * blockDim.x) {
            int r, c;
            if (a.tile_map) {                        // 64 lanes = 8 rows x one 32-column panel (see ApplyArgs.tile_map)
                const uint32_t it = e >> 6, rg = it / g4;
                r = (int)(rg * 8u + ((e >> 3) & 7u)); c = (int)((it - rg * g4) * 32u + (e & 7u) * 4u);
                if (r >= T || c >= fp) continue;
            } else {
                r = (int)(e / g4); c = (int)(e - (uint32_t)r * g4) * 4;
            }
            const float4 y = *reinterpret_cast<const float4*>(a.Y + (size_t)r * a.ldy + c);
            const float4 sc = *reinterpret_cast<const float4*>(a.bn + BN_SC * fp + c);
            const float4 sh = *reinterpret_cast<const float4*>(a.bn + BN_SH * fp + c);
            const float m = a.bt.row_m[r];
            float4 o;
            o.x = fmaxf(y.x * sc.x + sh.x, 0.0f) * m;
            o.y = fmaxf(y.y * sc.y + sh.y, 0.0f) * m;
            o.z = fmaxf(y.z * sc.z + sh.z, 0.0f) * m;
            o.w = fmaxf(y.w * sc.w + sh.w, 0.0f) * m;
            if (a.do_drop) {
                float ds[4];
                drop_scale4(seed, (uint64_t)r * fp + c, a.thr, a.inv_keep, ds);
                o.x *= ds[0]; o.y *= ds[1]; o.z *= ds[2]; o.w *= ds[3];
            }
            if (a.out) *reinterpret_cast<float4*>(a.out + (size_t)r * a.ldo + c) = o;
            if (a.planes.p) bx_store4(a.planes, r, c, o);
        }
    } else {
        // weighted sum over the views: a thread owns four adjacent output columns, the K view loads of a row are independent
        // 16-byte loads (view column ranges and the output pitch are multiples of 16)
        const uint32_t g4 = a.tile_map ? ((uint32_t)(a.ldo + 31) >> 5) : a.ldo / 4;
        const uint32_t total = a.tile_map ? (((uint32_t)T + 7) >> 3) * g4 * 64u : (uint32_t)T * g4;
        for (uint32_t e = blockIdx.x * blockDim.x + threadIdx.x; e < total; e += gridDim.x * blockDim.x) {
            int r, f;
            if (a.tile_map) {
                const uint32_t it = e >> 6, rg = it / g4;
                r = (int)(rg * 8u + ((e >> 3) & 7u)); f = (int)((it - rg * g4) * 32u + (e & 7u) * 4u);
                if (r >= T || f >= a.ldo) continue;
            } else {
                r = (int)(e / g4); f = (int)(e - (uint32_t)r * g4) * 4;
            }
            float4 y[EAGCN_MAX_VIEWS];
#pragma unroll
            for (int k = 0; k < EAGCN_MAX_VIEWS; ++k)
                if (k < a.vc.K) y[k] = *reinterpret_cast<const float4*>(a.Y + (size_t)r * a.ldy + a.vc.off[k] + f);
            float4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
            for (int k = 0; k < EAGCN_MAX_VIEWS; ++k)
                if (k < a.vc.K) {
                    const int cp = a.vc.off[k] + f;
                    const float4 sc = *reinterpret_cast<const float4*>(a.bn + BN_SC * fp + cp);
                    const float4 sh = *reinterpret_cast<const float4*>(a.bn + BN_SH * fp + cp);
                    const float4 w = *reinterpret_cast<const float4*>(a.colp + CP_AVEW * fp + cp);
                    float4 p;
                    p.x = fmaxf(y[k].x * sc.x + sh.x, 0.0f);
                    p.y = fmaxf(y[k].y * sc.y + sh.y, 0.0f);
                    p.z = fmaxf(y[k].z * sc.z + sh.z, 0.0f);
                    p.w = fmaxf(y[k].w * sc.w + sh.w, 0.0f);
                    if (a.do_drop) {
                        float ds[4];
                        drop_scale4(seed, (uint64_t)r * fp + cp, a.thr, a.inv_keep, ds);
                        p.x *= ds[0]; p.y *= ds[1]; p.z *= ds[2]; p.w *= ds[3];
                    }
                    acc.x += w.x * p.x; acc.y += w.y * p.y; acc.z += w.z * p.z; acc.w += w.w * p.w;
                }
            if (a.out) *reinterpret_cast<float4*>(a.out + (size_t)r * a.ldo + f) = acc;
            if (a.planes.p) bx_store4(a.planes, r, f, acc);
        }
    }
}

// ---- BatchNorm backward ----------------------------------------------------------------------------
struct BwdArgs {
    const int32_t* meta; int Tcap;       // (only what the kernel reads of the batch index: the argument block stays in SGPRs)
    const int32_t* row_mol;
    const float* row_m;
    ViewCols vc;
    int structure, fp, nvirt;
    const float* dxout; int ldo;
    ReadoutGrad rg;              // rg.dg != null: the upstream gradient is dg[row_mol[r]] (dxout unused)
    const float* dpad;           // [ldo] gradient of the common non-stored row, or null
    int dpad_views;              // dpad is [K][ldo]: one row per view (sampled dropout of the non-stored rows)
    const float* Y; int ldy;
    const float* bn;
    const float* colp;
    float* dH;                   // [T][fp]
    double* slab;                // [grid][fp][2]
    double* slab_da;             // [grid][MAX_VIEWS]
    int do_drop; uint32_t thr; float inv_keep; uint64_t seed; const uint64_t* seed_dev;
    double* zero; int nzero;     // fp64 words cleared on the way (the head's backward sums, for the next backward call)
    EdgeDrain drain;             // pending edge-gradient reduction of the layer above (eacc == nullptr: none)
    const float* cc;             // APPLY pass: [2][fp] means of dH and of dH * xhat (bn_bwd_finalize)
    int store_dh;                // first pass also stores dH (EAGCN_BWD_STORE_DH=1: the elementwise second pass of round 2 reads it)
};

// One workgroup owns BWD_ROWS consecutive-strided rows; a thread owns FOUR adjacent columns (one float4 per row
// and operand) and has all BWD_ROWS rows in flight at once: a single batch of independent 16-byte loads
// instead of a column loop of scalar ones.  View column ranges are multiples of 16, so the four columns share
// their view.
// Compile-time variants (structure, per-molecule upstream gradient, dropout): the general kernel carried every path at once
// -- 7.7 k instructions, scalar registers spilled to vector lanes, a few hundred exec-mask branches -- and was bound by
// that, not by memory.
// Two passes share this body, so that both form dH = upstream * dropout * relu' in exactly the same way:
//   APPLY = false: the column sums of dH and dH * xhat (fp64 slabs); dH itself is NOT stored;
//   APPLY = true (after bn_bwd_finalize): dH is formed again and dY' = sc * (dH - c1 - xhat * c2) is written.
// (Round 2 stored dH in the first pass and re-read it in an elementwise second one: one write and one read of T x Fp floats
//  more than re-forming it from Y and the upstream gradient, which the second pass reads anyway or gathers per molecule.)
template <bool WEIGHTED, bool DG, bool DROP, bool APPLY>
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(BwdArgs a) {
    __shared__ double da_s[EAGCN_MAX_VIEWS];
    const uint64_t seed = DROP ? (a.seed_dev ? *a.seed_dev : a.seed) : 0ull;
    const int fp = a.fp, T = min(a.meta[EAGCN_META_T], a.Tcap);
    const int rows = T + (APPLY ? 0 : a.nvirt);
    // the grid is sized for the row CAPACITY: only the first ceil(rows / BWD_ROWS) workgroups work (and write a
    // slab); bn_bwd_finalize derives the same count from the device-side row count
    const int nwg = max(1, min((int)gridDim.x, (rows + BWD_ROWS - 1) / BWD_ROWS));
    if (!APPLY && a.drain.eacc && blockIdx.x == 0 && blockIdx.y == 0) {
        // the layer above left its bond-type histograms in the shared accumulator slabs: reduce them into its d att.weight /
        // d self_r and leave the slabs zero for this layer's own edge gradients (agg_edge runs after this kernel)
        const int K = a.drain.K;
        for (int e = threadIdx.x; e < K * EDGE_SLAB; e += blockDim.x) {
            const int k = e / EDGE_SLAB, c = e - k * EDGE_SLAB;
            if (c > 256) continue;
            double v[EDGE_COPIES];
#pragma unroll
            for (int z = 0; z < EDGE_COPIES; ++z) v[z] = a.drain.eacc[((size_t)z * K + k) * EDGE_SLAB + c];
            double t = 0.0;
#pragma unroll
            for (int z = 0; z < EDGE_COPIES; ++z) { t += v[z]; a.drain.eacc[((size_t)z * K + k) * EDGE_SLAB + c] = 0.0; }
            float* dw = nullptr;
            float* dr = nullptr;
            int ch = 0;
#pragma unroll
            for (int q = 0; q < EAGCN_MAX_VIEWS; ++q)
                if (q == k) { dw = a.drain.datt_w[q]; dr = a.drain.dself_r[q]; ch = a.drain.channels[q]; }
            if (c == 256) {
                const double r = (double)a.drain.rsig[k];
                dr[0] = (float)(t * r * (1.0 - r));
            } else if (c >= 1 && c <= ch) {
                dw[c - 1] = (float)t;
            }
        }
    }
    if ((int)blockIdx.x >= nwg || (APPLY && T == 0)) return;
    if constexpr (!APPLY) {
        if (threadIdx.x < EAGCN_MAX_VIEWS) da_s[threadIdx.x] = 0.0;
        __syncthreads();
    }
    constexpr bool weighted = WEIGHTED;
    double da[EAGCN_MAX_VIEWS];
#pragma unroll
    for (int k = 0; k < EAGCN_MAX_VIEWS; ++k) da[k] = 0.0;
    // grid.y cuts the columns into chunks of 1024 (one pass of the workgroup): a wide layer (Fp = 6320 at the HIV widths) is
    // limited to a few hundred row blocks by the size of its partial slabs -- too few waves to stream at HBM rate
    const int c_lo = blockIdx.y * (int)blockDim.x * 4;
    for (int cp = c_lo + threadIdx.x * 4; cp < min(fp, c_lo + (int)blockDim.x * 4); cp += blockDim.x * 4) {
        const int k = col_view(a.vc, cp), f = cp - a.vc.off[k];
        const float4 sc = *reinterpret_cast<const float4*>(a.bn + BN_SC * fp + cp);
        const float4 sh = *reinterpret_cast<const float4*>(a.bn + BN_SH * fp + cp);
        const float4 mu = *reinterpret_cast<const float4*>(a.bn + BN_MU * fp + cp);
        const float4 iv = *reinterpret_cast<const float4*>(a.bn + BN_INV * fp + cp);
        const float4 aw = *reinterpret_cast<const float4*>(a.colp + CP_AVEW * fp + cp);
        float4 c1 = make_float4(0.f, 0.f, 0.f, 0.f), c2 = c1;
        if constexpr (APPLY) {
            c1 = *reinterpret_cast<const float4*>(a.cc + cp);
            c2 = *reinterpret_cast<const float4*>(a.cc + fp + cp);
        }
        const int cu = weighted ? f : cp;           // first column of the upstream gradient
        // ... and its exact column when that gradient is per molecule (ce4: the four columns are consecutive
        // exact columns, 16-byte aligned -> one load; otherwise each is looked up on its own)
        auto exact = [&](int c) {
            int eo = 0, po = 0, res = -1;
            bool done = false;
#pragma unroll
            for (int sg = 0; sg < EAGCN_MAX_SEGS; ++sg) {     // unrolled: constant indices into the argument block
                if (sg < a.rg.map.nseg && !done) {
                    if (c < po + a.rg.map.p[sg]) { res = (c - po < a.rg.map.w[sg]) ? eo + (c - po) : -1; done = true; }
                    eo += a.rg.map.w[sg];
                    po += a.rg.map.p[sg];
                }
            }
            return res;
        };
        int ce0 = -1, ce1 = -1, ce2 = -1, ce3 = -1;                // exact columns of the four packed columns: once per thread
        bool ce4 = false;
        if constexpr (DG) {
            ce0 = exact(cu); ce1 = exact(cu + 1); ce2 = exact(cu + 2); ce3 = exact(cu + 3);
            ce4 = ce0 >= 0 && ce3 == ce0 + 3 && (ce0 & 3) == 0 && (a.rg.F & 3) == 0 &&
                  (reinterpret_cast<uintptr_t>(a.rg.dg) & 15) == 0;
        }
        double s1[4] = {0.0, 0.0, 0.0, 0.0}, s2[4] = {0.0, 0.0, 0.0, 0.0}, dak = 0.0;
        for (int rb = blockIdx.x; rb < rows; rb += BWD_ROWS * nwg) {
            float4 yv[BWD_ROWS], upv[BWD_ROWS];
#pragma unroll
            for (int u = 0; u < BWD_ROWS; ++u) {
                const int r = rb + u * nwg;
                yv[u] = make_float4(0.f, 0.f, 0.f, 0.f);
                upv[u] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (r < T) {
                    yv[u] = *reinterpret_cast<const float4*>(a.Y + (size_t)r * a.ldy + cp);
                    if constexpr (DG) {
                        const int mol = a.row_mol[r];
                        const float* g = a.rg.dg + (size_t)mol * a.rg.F;
                        float4 v;
                        if (ce4) {
                            v = *reinterpret_cast<const float4*>(g + ce0);
                        } else {
                            v.x = ce0 >= 0 ? g[ce0] : 0.0f;
                            v.y = ce1 >= 0 ? g[ce1] : 0.0f;
                            v.z = ce2 >= 0 ? g[ce2] : 0.0f;
                            v.w = ce3 >= 0 ? g[ce3] : 0.0f;
                        }
                        if (a.rg.mode == 1) {
                            const float is = 1.0f / (float)a.rg.size[mol];
                            v.x *= is; v.y *= is; v.z *= is; v.w *= is;
                        }
                        upv[u] = v;
                    } else {
                        upv[u] = *reinterpret_cast<const float4*>(a.dxout + (size_t)r * a.ldo + cu);
                    }
                    if (!weighted) {
                        const float m = a.row_m[r];
                        upv[u].x *= m; upv[u].y *= m; upv[u].z *= m; upv[u].w *= m;
                    }
                } else if (r < rows) {                 // the one virtual row standing for all non-stored rows
                    upv[u] = *reinterpret_cast<const float4*>(a.dpad + (size_t)(a.dpad_views ? k : (r - T)) * a.ldo + cu);
                }
            }
#pragma unroll
            for (int u = 0; u < BWD_ROWS; ++u) {
                const int r = rb + u * nwg;
                if (r >= rows) continue;
                const float yy[4] = {yv[u].x, yv[u].y, yv[u].z, yv[u].w};
                const float uu[4] = {upv[u].x, upv[u].y, upv[u].z, upv[u].w};
                const float scv[4] = {sc.x, sc.y, sc.z, sc.w}, shv[4] = {sh.x, sh.y, sh.z, sh.w};
                const float muv[4] = {mu.x, mu.y, mu.z, mu.w}, ivv[4] = {iv.x, iv.y, iv.z, iv.w};
                const float awv[4] = {aw.x, aw.y, aw.z, aw.w};
                const float c1v[4] = {c1.x, c1.y, c1.z, c1.w}, c2v[4] = {c2.x, c2.y, c2.z, c2.w};
                float dh[4], dsv[4] = {1.0f, 1.0f, 1.0f, 1.0f};
                if (DROP && r < T) drop_scale4(seed, (uint64_t)r * fp + cp, a.thr, a.inv_keep, dsv);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float up = uu[j];
                    const float ds = dsv[j];
                    const float h = yy[j] * scv[j] + shv[j];
                    if (weighted) { if constexpr (!APPLY) dak += (double)(up * ds * fmaxf(h, 0.0f)); up *= awv[j]; }
                    dh[j] = h > 0.0f ? up * ds : 0.0f;
                    const float xh = (yy[j] - muv[j]) * ivv[j];
                    if constexpr (APPLY) {
                        dh[j] = scv[j] * (dh[j] - c1v[j] - xh * c2v[j]);
                    } else {
                        s1[j] += (double)dh[j];
                        s2[j] += (double)(dh[j] * xh);
                    }
                }
                if ((APPLY || a.store_dh) && r < T)
                    *reinterpret_cast<float4*>(a.dH + (size_t)r * fp + cp) = make_float4(dh[0], dh[1], dh[2], dh[3]);
            }
        }
        if constexpr (!APPLY) {
            double* sl = a.slab + ((size_t)blockIdx.x * fp + cp) * 2;
            *reinterpret_cast<double4*>(sl) = make_double4(s1[0], s2[0], s1[1], s2[1]);
            *reinterpret_cast<double4*>(sl + 4) = make_double4(s1[2], s2[2], s1[3], s2[3]);
            if (weighted) {
#pragma unroll
                for (int v = 0; v < EAGCN_MAX_VIEWS; ++v) da[v] += (v == k) ? dak : 0.0;
            }
        }
    }
    if (weighted && !APPLY) {
#pragma unroll
        for (int v = 0; v < EAGCN_MAX_VIEWS; ++v) {
            const double t = wave_sum(da[v]);
            if ((threadIdx.x & 63) == 0 && t != 0.0) atomicAdd(&da_s[v], t);
        }
        __syncthreads();
        if (threadIdx.x < EAGCN_MAX_VIEWS)
            a.slab_da[((size_t)blockIdx.y * gridDim.x + blockIdx.x) * EAGCN_MAX_VIEWS + threadIdx.x] = da_s[threadIdx.x];
    }
}

template <int L>
__global__ __launch_bounds__(256) void bn_bwd_finalize_kernel(const double* __restrict__ slab,
                                                               const double* __restrict__ slab_da, int nslab,
                                                               int fp, double M, int training,
                                                               const float* __restrict__ bn, ViewCols vc,
                                                               GradPtrs gp, float* __restrict__ cc, const int32_t* __restrict__ meta, int nvirt,
                                                               int batch_B, int da_chunks, int da_stride,
                                                               const double* __restrict__ gsum, double* __restrict__ zero, int nzero) {
    // fp64 words cleared on the way (the head's backward sums, for the next backward call; every reader -- the reduction in
    // front of this launch reads Graph_BN's sums -- is done by now)
    if (blockIdx.x == 0 && zero)
        for (int i = threadIdx.x; i < nzero; i += blockDim.x) zero[i] = 0.0;
    if (meta[EAGCN_META_NLOG] > 0) M = (double)batch_B * (double)meta[EAGCN_META_NLOG];
    nslab = max(1, min(nslab, (meta[EAGCN_META_T] + nvirt + BWD_ROWS - 1) / BWD_ROWS));   // slabs actually written
    const int cpr = blockIdx.x * (256 / L) + threadIdx.x / L, sl = threadIdx.x % L;
    const int cp = min(cpr, fp - 1);
    if (blockIdx.x == 0 && gp.dave_w) {                 // d ave.weight: 32 lanes per view over the row-partial slabs
        const int v = threadIdx.x >> 5, l = threadIdx.x & 31;
        double t = 0.0;
        if (v < vc.K)
            for (int y = 0; y < da_chunks; ++y)             // (column chunk y of the reduce grid, row block s)
                for (int s = l; s < nslab; s += 32) t += slab_da[((size_t)y * da_stride + s) * EAGCN_MAX_VIEWS + v];
#pragma unroll
        for (int o = 16; o > 0; o >>= 1) t += __shfl_xor(t, o);
        if (v < vc.K && l == 0) gp.dave_w[v] = (float)t;
    }
    double s1, s2;
    slab_sum<L>(slab, nslab, fp, cp, sl, s1, s2);
    if (cpr >= fp || sl != 0) return;
    {   // the means of dY = sc (dH - c1 - xhat c2) run over ALL rows of the BatchNorm: with sync-BatchNorm those of every rank
        double g1 = s1, g2 = s2;
        if (gsum) { g1 = gsum[2 * cp]; g2 = gsum[2 * cp + 1]; M = gsum[2 * fp]; }
        cc[cp] = training ? (float)(g1 / M) : 0.0f;
        cc[fp + cp] = training ? (float)(g2 / M) : 0.0f;
    }
    const int k = col_view(vc, cp), f = cp - vc.off[k];
    if (f < vc.width[k]) {
        gp.dgamma[k][f] = (float)s2;
        gp.dbeta[k][f] = (float)s1;
        // training: sum over ALL B*N rows of dY is identically zero (mean removal); eval: sc * sum(dH)
        gp.dbias[k][f] = training ? 0.0f : (float)((double)bn[BN_SC * fp + cp] * s1);
    }
}

__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(eagcn_batch bt, int fp, const float* __restrict__ Y, int ldy,
                                                            const float* __restrict__ bn,
                                                            const float* __restrict__ cc, float* __restrict__ dH) {
    const uint32_t g4 = fp / 4;
    const uint32_t total = (uint32_t)dev_rows(bt) * g4;
    for (uint32_t e = blockIdx.x * blockDim.x + threadIdx.x; e < total; e += gridDim.x * blockDim.x) {
        const int r = (int)(e / g4), c = (int)(e - (uint32_t)r * g4) * 4;
        const float4 y = *reinterpret_cast<const float4*>(Y + (size_t)r * ldy + c);
        float4 d = *reinterpret_cast<float4*>(dH + (size_t)r * fp + c);
        const float4 sc = *reinterpret_cast<const float4*>(bn + BN_SC * fp + c);
        const float4 mu = *reinterpret_cast<const float4*>(bn + BN_MU * fp + c);
        const float4 iv = *reinterpret_cast<const float4*>(bn + BN_INV * fp + c);
        const float4 c1 = *reinterpret_cast<const float4*>(cc + c);
        const float4 c2 = *reinterpret_cast<const float4*>(cc + fp + c);
        d.x = sc.x * (d.x - c1.x - (y.x - mu.x) * iv.x * c2.x);
        d.y = sc.y * (d.y - c1.y - (y.y - mu.y) * iv.y * c2.y);
        d.z = sc.z * (d.z - c1.z - (y.z - mu.z) * iv.z * c2.z);
        d.w = sc.w * (d.w - c1.w - (y.w - mu.w) * iv.w * c2.w);
        *reinterpret_cast<float4*>(dH + (size_t)r * fp + c) = d;
    }
}

// grid.x = ceil(ld_in*fp / 256) blocks for dW (one thread per element, sum over the split-K slabs) followed by
// ceil(K*EDGE_SLAB / 16) blocks for the edge-gradient partials (16 lanes per entry)
__global__ __launch_bounds__(256) void unpack_grads_kernel(GradPtrs gp, ParamPtrs pp, ViewCols vc, ColMapD in,
                                                            int ld_in, int fp, const float* __restrict__ dWcat,
                                                            int nsplit, size_t slab, double* __restrict__ datt,
                                                            int nedge, const float* __restrict__ rsig, int wblocks,
                                                            const int32_t* __restrict__ meta, int xk_G, int edge_drain, int bx_per, int bx_wide) {
    nedge = nedge < 0 ? -nedge : min(nedge, (meta[EAGCN_META_T] + 15) / 16);   // edge-gradient workgroups that had rows (< 0: all wrote)
    // split-K partials actually written: gemm.hip writes eff_splits of them; the plane GEMM (gemm_bx3.hip) passes its slab count
    // negated and writes bx3_used_splits() of them
    // (bx_per > 0: the weight gradient shared its launch with the dX product -- T x ld_in over fp -- and sized its chunks against it)
    // (tiles of the kernel that wrote them: 128 x 128, or 256 x 128 for gemm_bx3w.hip)
    const int bxm = bx_wide ? BX3W_BM : BX3_BM, bxn = bx_wide ? BX3W_BN : BX3_BN;
    nsplit = nsplit < 0 ? bx3_used_splits_wave(-nsplit, meta[EAGCN_META_T], ((ld_in + bxm - 1) / bxm) * ((fp + bxn - 1) / bxn),
                                          bx_per > 0 ? ((meta[EAGCN_META_T] + bxm - 1) / bxm) * ((ld_in + bxn - 1) / bxn) : 0, max(1, (fp + 31) >> 5), bx_per)
                        : max(1, min(nsplit, meta[EAGCN_META_T] >> 7));
    if ((int)blockIdx.x < wblocks) {
        // split-K slabs: FOUR lanes per group of four adjacent elements (one 16-byte load per slab: a wavefront covers 256 contiguous
        // bytes of four slabs), each lane adds every fourth slab (the first layer's weight gradient leaves gemm.hip as up to 146 slabs:
        // one thread per element was a chain of 37 dependent load rounds, 14 us at B = 1024; four lanes per SINGLE element, 64-byte
        // pieces: 15 us for the 14 slabs of configs[1]'s hidden layer)
        const int tid = blockIdx.x * blockDim.x + threadIdx.x;
        if (xk_G > 0) {
            const int e = tid;
            if (e >= ld_in * fp) return;
            const int ip = e / fp, cp = e % fp;
            const int k = col_view(vc, cp), f = cp - vc.off[k];
            const int fi = packed_to_exact(in, ip);
            if (fi < 0 || f >= vc.width[k]) return;
            // partial slabs of the XCD-local paired GEMM (gemm3.hip, kernels.h g3_plan): slab x exists iff segment x holds
            // k-steps of the dW product; summed in segment order (deterministic)
            const int T = meta[EAGCN_META_T];
            G3Plan pl;
            g3_plan((T + 63) / 64, (ld_in + 63) / 64, max(1, (fp + 15) / 16), ((ld_in + 63) / 64) * ((fp + 63) / 64),
                    max(1, (T + 15) / 16), xk_G, pl);
            float v[G3_XSEG];
#pragma unroll
            for (int x = 0; x < G3_XSEG; ++x) v[x] = pl.c[x + 1] > pl.c[x] ? dWcat[(size_t)x * slab + e] : 0.0f;
            float t = 0.0f;
#pragma unroll
            for (int x = 0; x < G3_XSEG; ++x) t += v[x];
            gp.dW[k][(size_t)fi * vc.width[k] + f] = T > 0 ? t : 0.0f;
            return;
        }
        const int sub = tid & 3;
        const int e = (tid >> 2) * 4;                                // (fp is a multiple of 16: the four elements share row and view)
        if (e >= ld_in * fp) return;
        const int ip = e / fp, cp = e % fp;
        const int k = col_view(vc, cp), f = cp - vc.off[k];
        const int fi = packed_to_exact(in, ip);
        if (fi < 0 || f >= vc.width[k]) return;
        float4 s0 = make_float4(0.f, 0.f, 0.f, 0.f), s1 = s0;
        int z = sub;
#pragma unroll 4
        for (; z + 4 < nsplit; z += 8) {
            const float4 a0 = *reinterpret_cast<const float4*>(dWcat + (size_t)z * slab + e);
            const float4 a1 = *reinterpret_cast<const float4*>(dWcat + (size_t)(z + 4) * slab + e);
            s0.x += a0.x; s0.y += a0.y; s0.z += a0.z; s0.w += a0.w;
            s1.x += a1.x; s1.y += a1.y; s1.z += a1.z; s1.w += a1.w;
        }
        if (z < nsplit) {
            const float4 a0 = *reinterpret_cast<const float4*>(dWcat + (size_t)z * slab + e);
            s0.x += a0.x; s0.y += a0.y; s0.z += a0.z; s0.w += a0.w;
        }
        float t[4] = {s0.x + s1.x, s0.y + s1.y, s0.z + s1.z, s0.w + s1.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) { t[j] += __shfl_xor(t[j], 1); t[j] += __shfl_xor(t[j], 2); }
        // lane j of the four stores element j
        const float tv = sub == 0 ? t[0] : sub == 1 ? t[1] : sub == 2 ? t[2] : t[3];
        if (f + sub < vc.width[k]) gp.dW[k][(size_t)fi * vc.width[k] + f + sub] = tv;
        return;
    }
    // edge-gradient partials [nedge][K][EDGE_SLAB]: entry c in 1..C_k -> d att_w[c-1]; entry 256 -> self term.
    // General relation vectors (pp.rel_vec[k]): the histogram is per bond CODE, the gradient per CHANNEL:
    //   d att_w[ch] = sum_codes h[code] * vec[code][ch]  -- entry ch of view k sums over all codes (rare path, not tuned)
    const int er = ((int)blockIdx.x - wblocks) * 16 + (threadIdx.x >> 4), sl = threadIdx.x & 15;
    const int tot = vc.K * EDGE_SLAB;
    const int e = min(er, tot - 1);
    const int k = e / EDGE_SLAB, c = e % EDGE_SLAB;
    const bool general = pp.rel_vec[k] != nullptr && c != 256;
    double t = 0.0;
    if (!general) {
        for (int z0 = sl; z0 < nedge; z0 += 16 * 8) {
            double v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int z = z0 + 16 * u;
                v[u] = z < nedge ? datt[(size_t)z * tot + e] : 0.0;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) t += v[u];
            if (edge_drain && er < tot) {                  // shared accumulator slabs: every element is read by exactly one lane
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int z = z0 + 16 * u;
                    if (z < nedge) datt[(size_t)z * tot + e] = 0.0;
                }
            }
        }
    } else if (c >= 1 && c <= pp.rel_c[k]) {
        const float* vec = pp.rel_vec[k];
        const int C = pp.rel_c[k], D = pp.channels[k];
        for (int z = sl; z < nedge; z += 16)
            for (int code = 1; code <= D; ++code)
                t += datt[(size_t)z * tot + k * EDGE_SLAB + code] * (double)vec[(size_t)(code - 1) * C + (c - 1)];
    }
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) t += __shfl_xor(t, o);
    if (er >= tot || sl != 0) return;
    if (c == 256) {
        const double r = (double)rsig[k];
        gp.dself_r[k][0] = (float)(t * r * (1.0 - r));
    } else if (c >= 1 && c <= (pp.rel_vec[k] ? pp.rel_c[k] : pp.channels[k])) {
        gp.datt_w[k][c - 1] = (float)t;
    }
}

// A1_k[b,i,j] = sigmoid(w_k[type]) * adj  -- the A_weight return value of layers.py:318 (stack of the
// per-view A1 of layers.py:83), materialised only when a caller asks for it.
__global__ __launch_bounds__(256) void attention_dense_kernel(eagcn_batch bt, ParamPtrs pp, float* __restrict__ out) {
    const size_t total = (size_t)bt.K * bt.B * bt.N * bt.N;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total;
         e += (size_t)gridDim.x * blockDim.x) {
        const int j = (int)(e % bt.N);
        const size_t kbi = e / bt.N;                 // (k*B + b)*N + i
        const int k = (int)(kbi / ((size_t)bt.B * bt.N));
        const size_t bi = kbi - (size_t)k * bt.B * bt.N;      // b*N + i: rows without bonds have no valid code row
        const uint32_t c = bt.deg_bn[bi] > 0 ? bt.code[kbi * bt.ldc + j] : 0u;
        out[e] = (c >= 1 && (int)c <= pp.channels[k]) ? sigmoidf_(att_logit(pp, k, (int)c)) : 0.0f;
    }
}

// ---- scratch carving -------------------------------------------------------------------------------
struct Carver {
    char* base;
    size_t off = 0;
    explicit Carver(void* p) : base((char*)p) {}
    template <typename T>
    T* take(size_t n) {
        T* p = base ? (T*)(base + off) : nullptr;
        off = align256(off + n * sizeof(T));
        return p;
    }
};

struct LayerDims {
    ViewCols vc;
    int fp, ld_in, fin, ldo, gx, gxb, nsplit;
    int gslab;                   // capacity of the forward BatchNorm partial slabs (the aggregation kernels' grids)
    size_t wslab;
    size_t wpslab, wtpslab;      // plane images (bx3.h) of Wcat [ld_in rows][fp] and WcatT [fp rows][ld_in]
    int np;                      // 3 / 1: this layer's products run from bf16 operand planes (gemm_bx3.hip); 0: fp32 operands
    int bx_splits;               // ... and its weight gradient leaves as this many k-chunk slabs
};
static LayerDims layer_dims(const eagcn_batch* b, const eagcn_layer_params* p) {
    LayerDims d;
    d.vc = view_cols(p);
    d.fp = d.vc.off[p->K];
    d.ld_in = layout_ld(&p->in);
    d.fin = layout_width(&p->in);
    d.ldo = p->structure == EAGCN_STRUCT_CONCATE ? d.fp : pad16(p->width[0]);
    d.gx = agg_grid_x(b);
    d.gslab = std::max(d.gx, lagg_use(b, 0, false, &d.vc) ? lagg_slabs(b) : 0);
    // row-partial slabs of the BatchNorm backward: 7 rows per workgroup, at most 2048 workgroups and at most
    // 32 MB of fp64 partials (wide layers: Fp = 6320 -> 331 workgroups).  Fewer, longer workgroups were measured
    // slower (the kernel is bound by its instruction stream and one memory round trip per 7-row batch, not by the
    // partial slabs: cap 512 -> +4 us, 192 -> +22 us at the Tox21 shape)
    {
        const long by_rows = cdiv(b->T + 1, BWD_ROWS);
        const long by_bytes = std::max<long>(64, (32L << 20) / ((long)d.fp * 16));
        d.gxb = (int)std::max<long>(1, std::min<long>(std::min<long>(by_rows, 2048), by_bytes));
    }
    const int tiles = cdiv(d.ld_in, 64) * cdiv(d.fp, 64);
    d.nsplit = std::max(1, std::min(std::max(1, 1024 / tiles), cdiv(std::max(b->T, 1), 128)));
    d.wslab = (size_t)d.ld_in * d.fp;
    d.wpslab = bx_plane_elems(d.ld_in, d.fp);
    d.wtpslab = bx_plane_elems(d.fp, d.ld_in);
    // plane GEMMs (gemm modes 3 / 4): the hidden layers (the 24-feature first layer stays on the fp32 kernels of gemm.hip)
    d.np = (d.ld_in >= 128 && (d.ld_in & 15) == 0 && (d.fp & 15) == 0) ? gemm_planes() : 0;
    // slab capacity of the weight gradient's k-chunks (how many are used is decided on the device from the actual row count:
    // bx3.h bx3_used_splits)
    // (at most 64 slabs, but never k-chunks beyond the 4096 rows gemm_bx3.hip's accumulation chains are bounded to: T > 262144 rows
    //  takes more slabs instead of longer chunks; < 1024 = the key packing of bx3_used_splits_wave)
    d.bx_splits = std::max(1, std::min(std::min(1023, std::max(64, cdiv(std::max(b->T, 1), 4096))), std::max(cdiv(std::max(b->T, 1), 4096), cdiv(std::max(b->T, 1), 256))));
    return d;
}

// Which layers run their products on the wave-autonomous balanced kernel (gemm3.hip): the hidden layers.  The first
// layer (24 atom features) is a different animal: its forward product has two k-steps per tile and its weight gradient
// seven 64x64 tiles with thousands of k-steps each -- a balanced cut would make every wave hand a partial tile to a
// single owner per tile (measured: 0.2 ms instead of 0.02 ms) -- it stays on the workgroup-tiled kernels (gemm.hip).
static bool gemm3_layer(int ld_in) {
    static const bool on = [] { const char* v = getenv("EAGCN_NO_GEMM3"); return !(v && v[0] == '1'); }();
    return on && ld_in >= 128 && gemm_mode() != 2;       // (the bf16 mode runs on the LDS-tiled kernels of gemm.hip)
}

struct Packed { float *Wcat, *WcatT, *colp, *sig, *rsig; uint16_t *Wp, *WTp; };   // Wp / WTp: [np][ld_in * fp] operand planes
struct FwdScratch { void* gws; double* eacc; float *Wcat, *WcatT, *colp, *sig, *rsig; uint16_t *Wp, *WTp; double* stats; double* gsum;
                    uint16_t* xp; };           // xp: planes of x when the caller did not bring them (layer-level path)
static size_t carve_packed(void* base, const LayerDims& d, Packed* s) {
    Carver c(base);
    Packed t;
    t.Wcat = c.take<float>(d.wslab);
    t.WcatT = c.take<float>(d.wslab);
    t.colp = c.take<float>((size_t)CP_ROWS * d.fp);
    t.sig = c.take<float>(EAGCN_MAX_VIEWS * 256);
    t.rsig = c.take<float>(EAGCN_MAX_VIEWS);
    t.Wp = c.take<uint16_t>(d.np ? (size_t)d.np * d.wpslab : 1);
    t.WTp = c.take<uint16_t>(d.np ? (size_t)d.np * d.wtpslab : 1);
    if (s) *s = t;
    return c.off;
}
struct BwdScratch {
    void* gws;                   // GEMM hand-off workspace: FIRST in both carvings, so that every layer of a model and both
                                 // directions share one region (one flag clear per API call, kernels.h gemm3_clear_flags)
    double* eacc;                // edge-gradient accumulator slabs (kernels.h EDGE_COPIES): right behind it, zero between uses
    float *Wcat, *WcatT, *colp, *sig, *rsig, *dY, *dP, *cc, *dWcat;
    uint16_t *Wp, *WTp;
    double *slab, *slab_da, *datt, *gsum;
    uint16_t *dPp, *xp;          // planes of dP (written by the transposed aggregation) and of x when the caller did not bring them
};
static size_t carve_fwd(void* base, const eagcn_batch* b, const LayerDims& d, FwdScratch* s) {
    Carver c(base);
    FwdScratch t;
    t.gws = c.take<char>(gemm3_workspace_bytes());
    t.eacc = (double*)c.take<char>(edge_acc_bytes());        // (same place in both carvings: kernels.h EDGE_COPIES)
    t.Wcat = c.take<float>(d.wslab);
    t.WcatT = c.take<float>(d.wslab);
    t.colp = c.take<float>((size_t)CP_ROWS * d.fp);
    t.sig = c.take<float>(EAGCN_MAX_VIEWS * 256);
    t.rsig = c.take<float>(EAGCN_MAX_VIEWS);
    t.Wp = c.take<uint16_t>(d.np ? (size_t)d.np * d.wpslab : 1);
    t.WTp = c.take<uint16_t>(d.np ? (size_t)d.np * d.wtpslab : 1);
    t.stats = c.take<double>((size_t)d.gslab * d.fp * 2);
    t.gsum = c.take<double>((size_t)2 * d.fp + 8);
    t.xp = c.take<uint16_t>(d.np ? (size_t)d.np * bx_plane_elems(b->T, d.ld_in) : 1);
    if (s) *s = t;
    return c.off;
}
static size_t carve_bwd(void* base, const eagcn_batch* b, const LayerDims& d, BwdScratch* s) {
    Carver c(base);
    BwdScratch t;
    t.gws = c.take<char>(gemm3_workspace_bytes());
    t.eacc = (double*)c.take<char>(edge_acc_bytes());
    t.Wcat = c.take<float>(d.wslab);
    t.WcatT = c.take<float>(d.wslab);
    t.colp = c.take<float>((size_t)CP_ROWS * d.fp);
    t.sig = c.take<float>(EAGCN_MAX_VIEWS * 256);
    t.rsig = c.take<float>(EAGCN_MAX_VIEWS);
    t.Wp = c.take<uint16_t>(d.np ? (size_t)d.np * d.wpslab : 1);
    t.WTp = c.take<uint16_t>(d.np ? (size_t)d.np * d.wtpslab : 1);
    t.dY = c.take<float>((size_t)std::max(b->T, 1) * d.fp);
    t.dP = c.take<float>((size_t)std::max(b->T, 1) * d.fp);
    t.cc = c.take<float>((size_t)2 * d.fp);
    t.dWcat = c.take<float>(d.wslab * std::max(std::max(d.nsplit, G3_XSEG), d.np ? d.bx_splits : 1));
    t.slab = c.take<double>((size_t)d.gxb * d.fp * 2);
    t.slab_da = c.take<double>((size_t)d.gxb * cdiv(d.fp, 1024) * EAGCN_MAX_VIEWS);
    t.datt = c.take<double>((size_t)edge_grid_x(b) * EAGCN_MAX_VIEWS * EDGE_SLAB);
    t.gsum = c.take<double>((size_t)2 * d.fp + 8);
    t.dPp = c.take<uint16_t>(d.np ? (size_t)d.np * bx_plane_elems(b->T, d.fp) : 1);
    t.xp = c.take<uint16_t>(d.np ? (size_t)d.np * bx_plane_elems(b->T, d.ld_in) : 1);
    if (s) *s = t;
    return c.off;
}

static int check_layer(const eagcn_batch* b, const eagcn_layer_params* p, const char* who) {
    EAGCN_CHECK_ARG(b && p, "%s: null argument", who);
    EAGCN_CHECK_ARG(p->K >= 1 && p->K <= EAGCN_MAX_VIEWS && p->K == b->K,
                    "%s: layer has %d views, batch index has %d", who, p->K, b->K);
    EAGCN_CHECK_ARG(p->structure == EAGCN_STRUCT_CONCATE || p->structure == EAGCN_STRUCT_WEIGHTED,
                    "%s: unknown structure %d", who, p->structure);
    EAGCN_CHECK_ARG(p->in.nseg >= 1 && p->in.nseg <= EAGCN_MAX_SEGS, "%s: bad input layout", who);
    for (int s = 0; s < p->in.nseg; ++s)
        EAGCN_CHECK_ARG(p->in.width[s] >= 1 && p->in.pad[s] >= p->in.width[s] && (p->in.pad[s] % 4) == 0,
                        "%s: input segment %d: width %d pad %d (pad must be a multiple of 4 >= width)", who, s,
                        p->in.width[s], p->in.pad[s]);
    for (int k = 0; k < p->K; ++k) {
        EAGCN_CHECK_ARG(p->width[k] >= 1, "%s: view %d has width %d", who, k, p->width[k]);
        EAGCN_CHECK_ARG(p->att_w[k] && p->self_r[k] && p->W[k] && p->bias[k] && p->gamma[k] && p->beta[k] &&
                            p->run_mean[k] && p->run_var[k], "%s: view %d has a null parameter", who, k);
        if (p->structure == EAGCN_STRUCT_WEIGHTED)
            EAGCN_CHECK_ARG(p->width[k] == p->width[0], "%s: Weighted_sum needs equal view widths", who);
    }
    if (p->structure == EAGCN_STRUCT_WEIGHTED) EAGCN_CHECK_ARG(p->ave_w, "%s: Weighted_sum needs ave_w", who);
    EAGCN_CHECK_ARG(p->dropout >= 0.0f && p->dropout < 1.0f, "%s: dropout %f out of [0,1)", who, p->dropout);
    // the elementwise BatchNorm passes index (row, 4-column group) pairs with 32 bits
    EAGCN_CHECK_ARG((uint64_t)std::max(b->T, 0) * (uint64_t)view_cols(p).off[p->K] / 4 < (1ull << 32),
                    "%s: %d rows x %d packed columns exceed the 32-bit element index of the BatchNorm passes", who, b->T,
                    view_cols(p).off[p->K]);
    return EAGCN_OK;
}

static ParamPtrs param_ptrs(const eagcn_batch* b, const eagcn_layer_params* p) {
    ParamPtrs pp;
    memset(&pp, 0, sizeof(pp));
    for (int k = 0; k < p->K; ++k) {
        pp.att_w[k] = p->att_w[k]; pp.self_r[k] = p->self_r[k]; pp.W[k] = p->W[k]; pp.bias[k] = p->bias[k];
        pp.gamma[k] = p->gamma[k]; pp.beta[k] = p->beta[k]; pp.run_mean[k] = p->run_mean[k];
        pp.run_var[k] = p->run_var[k]; pp.channels[k] = b->channels[k];
        pp.rel_vec[k] = b->rel_vec[k]; pp.rel_c[k] = b->rel_c[k];
    }
    pp.ave_w = p->structure == EAGCN_STRUCT_WEIGHTED ? p->ave_w : nullptr;
    return pp;
}

static inline int ew_grid(size_t n) { return (int)std::max<size_t>(1, std::min<size_t>((n + 255) / 256, 2048)); }

}  // namespace eagcn

static int apply_launch_(const eagcn_batch* b, const eagcn_layer_params* p, const eagcn_layer_bufs* w, const eagcn::LayerDims& d,
                         const float* colp, hipStream_t s, bool planes_only = false);
#define apply_launch apply_launch_

using namespace eagcn;

extern "C" int eagcn_pad16(int w) { return pad16(w); }
extern "C" int eagcn_layer_fp(const eagcn_layer_params* p) { return view_cols(p).off[p->K]; }
extern "C" int eagcn_layer_out_ld(const eagcn_layer_params* p) {
    return p->structure == EAGCN_STRUCT_CONCATE ? eagcn_layer_fp(p) : pad16(p->width[0]);
}
extern "C" size_t eagcn_layer_packed_bytes(const eagcn_batch* b, const eagcn_layer_params* p) {
    LayerDims d = layer_dims(b, p);
    return carve_packed(nullptr, d, nullptr);
}
extern "C" size_t eagcn_layer_fwd_scratch_bytes(const eagcn_batch* b, const eagcn_layer_params* p) {
    LayerDims d = layer_dims(b, p);
    return carve_fwd(nullptr, b, d, nullptr);
}
extern "C" size_t eagcn_layer_bwd_scratch_bytes(const eagcn_batch* b, const eagcn_layer_params* p) {
    LayerDims d = layer_dims(b, p);
    return carve_bwd(nullptr, b, d, nullptr);
}

extern "C" int eagcn_layer_forward(const eagcn_batch* b, const eagcn_layer_params* p, const eagcn_layer_bufs* w,
                                   void* stream) {
    EAGCN_CHECK_GEMM3("eagcn_layer_forward");
    if (w && w->scratch && w->scratch_bytes >= gemm3_workspace_bytes()) {
        int rc = gemm3_clear_flags(w->scratch, w->scratch_bytes, (hipStream_t)stream);
        if (rc) return rc;
    }
    return layer_forward_impl(b, p, w, stream, false);
}

// parameters of up to four layers re-laid into their `packed` blocks by ONE launch (model engine)
int eagcn::pack_params_all(const eagcn_batch* b, const eagcn_layer_params* const* ps, void* const* packed,
                           const size_t* packed_bytes, int n, void* stream, const ZeroJob* zj, const HandOff* ho) {
    hipStream_t s = (hipStream_t)stream;
    EAGCN_CHECK_ARG(n >= 1 && n <= 4, "pack_params_all: 1..4 layers");
    PackJob jobs[4];
    size_t wmax = 0;
    for (int l = 0; l < 4; ++l) {
        const int ll = l < n ? l : n - 1;
        int rc = check_layer(b, ps[ll], "eagcn_model_forward");
        if (rc) return rc;
        const LayerDims d = layer_dims(b, ps[ll]);
        Packed pk;
        const size_t pneed = carve_packed(packed[ll], d, &pk);
        if (pneed > packed_bytes[ll]) {
            set_error("pack_params_all: packed buffer of layer %d too small (%zu < %zu)", ll, packed_bytes[ll], pneed);
            return EAGCN_ERR_SCRATCH;
        }
        jobs[l].pp = param_ptrs(b, ps[ll]); jobs[l].vc = d.vc; jobs[l].in = make_colmap(&ps[ll]->in);
        jobs[l].ld_in = d.ld_in; jobs[l].fp = d.fp;
        jobs[l].Wcat = pk.Wcat; jobs[l].WcatT = pk.WcatT; jobs[l].colp = pk.colp; jobs[l].sig = pk.sig; jobs[l].rsig = pk.rsig;
        jobs[l].wp = BxOut{d.np ? pk.Wp : nullptr, d.wpslab, d.np, d.ld_in};
        jobs[l].wtp = BxOut{d.np ? pk.WTp : nullptr, d.wtpslab, d.np, d.fp};
        if (l < n) wmax = std::max(wmax, d.wslab);
    }
    PackJobs pj{jobs[0], jobs[1], jobs[2], jobs[3]};
    ProfScope ps_(PROF_PACK, s);
    const ZeroJob none{nullptr, 0, nullptr, 0};
    const HandOff noho{nullptr, nullptr, nullptr, 0};
    pack_params_multi_kernel<<<dim3(ew_grid(wmax), n), 256, 0, s>>>(pj, zj ? *zj : none, ho ? *ho : noho);
    EAGCN_LAUNCH_CHECK();
    return EAGCN_OK;
}

int eagcn::layer_forward_impl(const eagcn_batch* b, const eagcn_layer_params* p, const eagcn_layer_bufs* w,
                              void* stream, bool prepacked, bool skip_apply, bool planes_only) {
    hipStream_t s = (hipStream_t)stream;
    int rc = check_layer(b, p, "eagcn_layer_forward");
    if (rc) return rc;
    EAGCN_CHECK_ARG(w && w->bn && w->xout && w->pad_row && w->scratch, "eagcn_layer_forward: null buffer");
    // (w->x may be null when the layer below wrote plane images only: every fp32 fallback below then refuses)
    EAGCN_CHECK_ARG(b->T == 0 || ((w->x || w->x_planes) && w->P && w->Y && w->rscale), "eagcn_layer_forward: null activation buffer");
    EAGCN_CHECK_ARG(!planes_only || (w->xout_planes && gemm_planes()), "eagcn_layer_forward: planes-only output without plane images");
    const LayerDims d = layer_dims(b, p);
    FwdScratch sc;
    const size_t need = carve_fwd(w->scratch, b, d, &sc);
    if (need > w->scratch_bytes) {
        set_error("eagcn_layer_forward: scratch too small (%zu < %zu)", w->scratch_bytes, need);
        return EAGCN_ERR_SCRATCH;
    }
    const ParamPtrs pp = param_ptrs(b, p);
    const ColMapD in = make_colmap(&p->in);
    if (w->packed) {
        Packed pk;
        const size_t pneed = carve_packed(w->packed, d, &pk);
        if (pneed > w->packed_bytes) {
            set_error("eagcn_layer_forward: packed buffer too small (%zu < %zu)", w->packed_bytes, pneed);
            return EAGCN_ERR_SCRATCH;
        }
        sc.Wcat = pk.Wcat; sc.WcatT = pk.WcatT; sc.colp = pk.colp; sc.sig = pk.sig; sc.rsig = pk.rsig;
        sc.Wp = pk.Wp; sc.WTp = pk.WTp;
    }
    EAGCN_CHECK_ARG(!prepacked || w->packed, "eagcn_layer_forward: prepacked parameters need the packed block");
    if (!prepacked) {
        ProfScope ps(PROF_PACK, s);
        pack_params_kernel<<<ew_grid(d.wslab), 256, 0, s>>>(pp, d.vc, in, d.ld_in, d.fp, sc.Wcat, sc.WcatT, sc.colp, sc.sig, sc.rsig,
                                                            BxOut{d.np ? sc.Wp : nullptr, d.wpslab, d.np, d.ld_in},
                                                            BxOut{d.np ? sc.WTp : nullptr, d.wtpslab, d.np, d.fp});
    }
    EAGCN_LAUNCH_CHECK();
    // algorithmic flops of the flat transform: exact widths, packed rows
    double fsum = 0.0;
    for (int k = 0; k < p->K; ++k) fsum += p->width[k];
    const double gemm_work = 2.0 * (double)b->T * (double)d.fin * fsum;
    int nslab = 0, tiles_per_wg = agg_ksplit(b) ? 1 : 4;
    if (b->T > 0) {
        // P = X.[W_1|..|W_K]: NT form on the pre-transposed weight (wave-autonomous balanced kernel, gemm3.hip); operands
        // that are not 16-byte aligned fall back to the workgroup-tiled kernel
        GemmDesc g3{0, 1, b->T, d.fp, d.ld_in, w->x, d.ld_in, sc.WcatT, d.ld_in, w->P, d.fp, 1, 0, gemm_work};
        g3.M_dev = b->meta + EAGCN_META_T;
        BxProb bp;
        memset(&bp, 0, sizeof(bp));
        if (d.np) {
            // the same product from bf16 operand planes (gemm_bx3.hip): x planes from the layer below, or split here
            const size_t xstride = bx_plane_elems(b->T, d.ld_in);
            const uint16_t* xp = w->x_planes;
            if (!xp) {
                EAGCN_CHECK_ARG(w->x, "eagcn_layer_forward: neither an fp32 input nor its plane images");
                rc = launch_bx3_split(w->x, b->T, b->meta + EAGCN_META_T, d.ld_in, sc.xp, xstride, b->T, d.np, s);
                if (rc) return rc;
                xp = sc.xp;
            }
            bp.A = BxPlanes{xp, xstride, d.ld_in, b->T}; bp.B = BxPlanes{sc.WTp, d.wtpslab, d.ld_in, d.fp}; bp.C = w->P; bp.ldc = d.fp;
            bp.M = b->T; bp.N = d.fp; bp.K = d.ld_in; bp.M_dev = b->meta + EAGCN_META_T; bp.tn = 0; bp.splits = 1;
        }
        if (d.np && bx3_ok(bp)) {
            rc = launch_bx3(bp, nullptr, d.np, s, gemm_work, PROF_GEMM, bx3_pick_wide(bp, nullptr, b->t_hint));
        } else if (!w->x) {
            set_error("eagcn_layer_forward: the input exists as plane images only and the plane product does not apply");
            return EAGCN_ERR_ARG;
        } else if (gemm3_layer(d.ld_in) && gemm3_ok(g3)) {
            rc = launch_gemm3(g3, nullptr, sc.gws, gemm3_workspace_bytes(), s);
        } else {
            GemmDesc g{0, 0, b->T, d.fp, d.ld_in, w->x, d.ld_in, sc.Wcat, d.fp, w->P, d.fp, 1, 0, gemm_work};
            g.M_dev = b->meta + EAGCN_META_T;
            rc = launch_gemm(g, s);
        }
        if (rc) return rc;
        {
            AggArgs a;
            a.bt = *b; a.vc = d.vc; a.src = w->P; a.lds = d.fp; a.dst = w->Y; a.ldd = d.fp;
            a.sig = sc.sig; a.rsig = sc.rsig; a.rscale = w->rscale; a.stats = sc.stats; a.nchunk = 1;
            if (lagg_use(b, 0, false, &d.vc)) {                // LDS-staged bond-list aggregation (lagg.hip): one slab per row block
                rc = launch_lagg_fwd(a, s);
                nslab = lagg_slabs(b);
                tiles_per_wg = -1;
            } else {
                rc = launch_agg(a, false, s);
                nslab = d.gx;
            }
            if (rc) return rc;
        }
    }
    const double M = (double)b->B * (double)b->N;
    {
    ProfScope psbn(PROF_BN, s);
    if (w->stats_hook && p->training) {
        // sync-BatchNorm: this rank's sums -> one vector, summed across the ranks by the caller's hook, finalize from that
        if (nslab > 64) bn_stats_sum_kernel<64><<<cdiv(d.fp, 4), 256, 0, s>>>(sc.stats, nslab, d.fp, M, sc.gsum, b->meta, 1, tiles_per_wg, 0, b->B);
        else bn_stats_sum_kernel<16><<<cdiv(d.fp, 16), 256, 0, s>>>(sc.stats, nslab, d.fp, M, sc.gsum, b->meta, 1, tiles_per_wg, 0, b->B);
        EAGCN_LAUNCH_CHECK();
        if (w->stats_hook(sc.gsum, 2 * d.fp + 1, stream, w->stats_user)) {
            set_error("eagcn_layer_forward: the sync-BatchNorm all-reduce hook failed");
            return EAGCN_ERR_HIP;
        }
        bn_finalize_kernel<16><<<cdiv(d.fp, 16), 256, 0, s>>>(sc.gsum, 1, d.fp, M, p->training, p->bn_eps, p->bn_momentum, sc.colp,
                                                                pp, d.vc, w->bn, b->meta, 0, b->B, sc.gsum + 2 * d.fp);
    } else if (nslab > 64)
        bn_finalize_kernel<64><<<cdiv(d.fp, 4), 256, 0, s>>>(sc.stats, nslab, d.fp, M, p->training, p->bn_eps,
                                                               p->bn_momentum, sc.colp, pp, d.vc, w->bn, b->meta, tiles_per_wg, b->B, nullptr);
    else
        bn_finalize_kernel<16><<<cdiv(d.fp, 16), 256, 0, s>>>(sc.stats, nslab, d.fp, M, p->training, p->bn_eps,
                                                                p->bn_momentum, sc.colp, pp, d.vc, w->bn, b->meta, tiles_per_wg, b->B, nullptr);
    EAGCN_LAUNCH_CHECK();
    }
    if (skip_apply) return EAGCN_OK;
    return apply_launch(b, p, w, d, sc.colp, s, planes_only);
}

bool eagcn::layer_reads_planes_only(const eagcn_batch* b, const eagcn_layer_params* p, bool aux_stream) {
    static const bool on = [] { const char* v = getenv("EAGCN_PLANES_ONLY"); return !(v && v[0] == '0'); }();
    if (!on || aux_stream || !b || !p || b->T <= 0) return false;
    const LayerDims d = layer_dims(b, p);
    if (!d.np) return false;
    // the three products exactly as layer_forward_impl / layer_backward_impl describe them (bx3_ok looks at alignment, strides
    // and extents: the operands below stand for the 256-byte aligned pieces those functions carve)
    const uint16_t* pl = reinterpret_cast<const uint16_t*>(static_cast<uintptr_t>(4096));
    float* fo = reinterpret_cast<float*>(static_cast<uintptr_t>(4096));
    const size_t xstride = bx_plane_elems(b->T, d.ld_in), pstride = bx_plane_elems(b->T, d.fp);
    BxProb bp, bx, bw;
    memset(&bp, 0, sizeof(bp)); memset(&bx, 0, sizeof(bx)); memset(&bw, 0, sizeof(bw));
    bp.A = BxPlanes{pl, xstride, d.ld_in, b->T}; bp.B = BxPlanes{pl, d.wtpslab, d.ld_in, d.fp}; bp.C = fo; bp.ldc = d.fp;
    bp.M = b->T; bp.N = d.fp; bp.K = d.ld_in; bp.M_dev = b->meta + EAGCN_META_T; bp.tn = 0; bp.splits = 1;
    bx.A = BxPlanes{pl, pstride, d.fp, b->T}; bx.B = BxPlanes{pl, d.wpslab, d.fp, d.ld_in}; bx.C = fo; bx.ldc = d.ld_in;
    bx.M = b->T; bx.N = d.ld_in; bx.K = d.fp; bx.M_dev = b->meta + EAGCN_META_T; bx.tn = 0; bx.splits = 1;
    bw.A = BxPlanes{pl, xstride, d.ld_in, b->T}; bw.B = BxPlanes{pl, pstride, d.fp, b->T};
    bw.C = fo; bw.ldc = d.fp; bw.M = d.ld_in; bw.N = d.fp; bw.K = b->T; bw.K_dev = b->meta + EAGCN_META_T; bw.tn = 1;
    bw.splits = d.bx_splits; bw.slab = d.wslab;
    return bx3_ok(bp) && bx3_ok(bw) && bx3_ok(bx);
}

int eagcn::layer_apply_impl(const eagcn_batch* b, const eagcn_layer_params* p, const eagcn_layer_bufs* w, void* stream) {
    EAGCN_CHECK_ARG(b && p && w && w->bn && w->xout && w->pad_row && w->packed, "layer_apply: null buffer");
    const LayerDims d = layer_dims(b, p);
    Packed pk;
    EAGCN_CHECK_ARG(carve_packed(w->packed, d, &pk) <= w->packed_bytes, "layer_apply: packed buffer too small");
    return apply_launch(b, p, w, d, pk.colp, (hipStream_t)stream);
}

static int apply_launch_(const eagcn_batch* b, const eagcn_layer_params* p, const eagcn_layer_bufs* w, const LayerDims& d,
                         const float* colp, hipStream_t s, bool planes_only) {
    ProfScope psbn(PROF_BN, s);
    ApplyArgs aa;
    aa.bt = *b; aa.vc = d.vc; aa.structure = p->structure; aa.fp = d.fp; aa.Y = w->Y; aa.ldy = d.fp;
    aa.bn = w->bn; aa.colp = colp; aa.out = w->xout; aa.ldo = d.ldo; aa.pad_row = w->pad_row;
    aa.do_drop = (p->training && p->dropout > 0.0f) ? 1 : 0;
    aa.thr = (uint32_t)std::min(4294967295.0, (double)p->dropout * 4294967296.0);
    aa.inv_keep = 1.0f / (1.0f - p->dropout);
    aa.seed = p->seed;
    aa.seed_dev = p->seed_dev;
    aa.planes = BxOut{gemm_planes() ? w->xout_planes : nullptr, bx_plane_elems(b->T, d.ldo), gemm_planes(), b->T};
    if (planes_only && aa.planes.p) aa.out = nullptr;        // (the only reader is the layer above, through the planes)
    static const int map_env = [] { const char* v = getenv("EAGCN_BN_APPLY_MAP"); return v ? atoi(v) : -1; }();
    aa.tile_map = map_env >= 0 ? map_env : (aa.planes.p ? 1 : 0);
    const size_t nthr = aa.tile_map ? (size_t)cdiv(std::max(b->T, 1), 8) * cdiv(d.ldo, 32) * 64 : (size_t)std::max(b->T, 1) * d.ldo / 4;
    bn_apply_kernel<<<ew_grid(nthr), 256, 0, s>>>(aa);
    EAGCN_LAUNCH_CHECK();
    return EAGCN_OK;
}

// Second pass of the BatchNorm backward: re-form dH in the reduction kernel's body (APPLY) or read the dH the first pass stored
// (elementwise bn_bwd_apply_kernel).  Weighted_sum: the upstream gradient is K times narrower than dH, re-forming it saves a
// write and a read of T x Fp floats (HIV widths: 9.74 -> 9.53 ms per step).  Concate: the upstream gradient is as wide as dH
// and the plain elementwise pass is the lighter kernel (B = 256: 0.456 vs 0.459 ms; B = 1024: equal).
// EAGCN_BWD_STORE_DH=1 / 0 forces one or the other.
// the elementwise second pass of the BatchNorm backward disappears into lagg.hip's staging whenever that kernel consumes dY' (it is
// the ONLY consumer then: transposed aggregation and edge gradients in one launch); EAGCN_LAGG_FUSE_BN=0 keeps the pass
static bool lagg_fuses_bn() {
    static const bool on = [] { const char* v = getenv("EAGCN_LAGG_FUSE_BN"); return !(v && v[0] == '0'); }();
    return on;
}
// Weighted_sum layers (round 6): lagg.hip re-forms dH from the K-times-narrower upstream gradient in its staging, so the second pass
// (a read of Y' and a write of T x K F floats) and the read of dY' behind it disappear; EAGCN_LAGG_WFUSE=0 keeps the pass
// (lagg_wfuse(): lagg.hip, where the policy lives)
static bool bn_bwd_two_pass(bool weighted) {
    static const int force = [] { const char* v = getenv("EAGCN_BWD_STORE_DH"); return v ? (v[0] == '1' ? 1 : 0) : -1; }();
    return force < 0 ? weighted : force == 0;
}
// the two passes of bn_bwd_reduce_kernel (apply = false: sums, apply = true: dY') in their compile-time variants
static void launch_bn_bwd_pass(bool apply, bool wt, bool dg, bool dr, dim3 grid, const BwdArgs& ba, hipStream_t s) {
#define EAGCN_BWD(W, G, D) do { if (apply) bn_bwd_reduce_kernel<W, G, D, true><<<grid, 256, 0, s>>>(ba); \
                                else bn_bwd_reduce_kernel<W, G, D, false><<<grid, 256, 0, s>>>(ba); } while (0)
    if (wt) { if (dg) { if (dr) EAGCN_BWD(true, true, true); else EAGCN_BWD(true, true, false); }
              else    { if (dr) EAGCN_BWD(true, false, true); else EAGCN_BWD(true, false, false); } }
    else    { if (dg) { if (dr) EAGCN_BWD(false, true, true); else EAGCN_BWD(false, true, false); }
              else    { if (dr) EAGCN_BWD(false, false, true); else EAGCN_BWD(false, false, false); } }
#undef EAGCN_BWD
}

extern "C" int eagcn_layer_backward(const eagcn_batch* b, const eagcn_layer_params* p, const eagcn_layer_bufs* w,
                                    const float* dxout, const float* dpad_row, float* dx,
                                    const eagcn_layer_grads* g, void* stream) {
    EAGCN_CHECK_GEMM3("eagcn_layer_backward");
    if (w && w->scratch && w->scratch_bytes >= gemm3_workspace_bytes()) {
        int rc = gemm3_clear_flags(w->scratch, w->scratch_bytes, (hipStream_t)stream);
        if (rc) return rc;
    }
    return layer_backward_impl(b, p, w, dxout, nullptr, dpad_row, dx, g, stream);
}

int eagcn::layer_backward_impl(const eagcn_batch* b, const eagcn_layer_params* p, const eagcn_layer_bufs* w,
                               const float* dxout, const ReadoutGrad* rg, const float* dpad_row, float* dx,
                               const eagcn_layer_grads* g, void* stream, bool dpad_views, const ZeroJob* zero_after,
                               const EdgeDrain* drain_in, EdgeDrain* drain_out) {
    if (drain_out) drain_out->eacc = nullptr;
    hipStream_t s = (hipStream_t)stream;
    int rc = check_layer(b, p, "eagcn_layer_backward");
    if (rc) return rc;
    EAGCN_CHECK_ARG(w && g && w->bn && w->scratch, "eagcn_layer_backward: null buffer");
    EAGCN_CHECK_ARG(b->T == 0 || ((dxout || rg) && (w->x || w->x_planes) && w->P && w->Y && w->rscale), "eagcn_layer_backward: null activation buffer");
    for (int k = 0; k < p->K; ++k)
        EAGCN_CHECK_ARG(g->dW[k] && g->dbias[k] && g->dgamma[k] && g->dbeta[k] && g->datt_w[k] && g->dself_r[k],
                        "eagcn_layer_backward: view %d has a null gradient buffer", k);
    const LayerDims d = layer_dims(b, p);
    BwdScratch sc;
    const size_t need = carve_bwd(w->scratch, b, d, &sc);
    if (need > w->scratch_bytes) {
        set_error("eagcn_layer_backward: scratch too small (%zu < %zu)", w->scratch_bytes, need);
        return EAGCN_ERR_SCRATCH;
    }
    const ParamPtrs pp = param_ptrs(b, p);
    GradPtrs gp;
    memset(&gp, 0, sizeof(gp));
    for (int k = 0; k < p->K; ++k) {
        gp.dW[k] = g->dW[k]; gp.dbias[k] = g->dbias[k]; gp.dgamma[k] = g->dgamma[k]; gp.dbeta[k] = g->dbeta[k];
        gp.datt_w[k] = g->datt_w[k]; gp.dself_r[k] = g->dself_r[k];
    }
    gp.dave_w = p->structure == EAGCN_STRUCT_WEIGHTED ? g->dave_w : nullptr;
    const ColMapD in = make_colmap(&p->in);
    if (w->packed) {          // parameters were re-laid by the forward call and kept
        Packed pk;
        const size_t pneed = carve_packed(w->packed, d, &pk);
        if (pneed > w->packed_bytes) {
            set_error("eagcn_layer_backward: packed buffer too small (%zu < %zu)", w->packed_bytes, pneed);
            return EAGCN_ERR_SCRATCH;
        }
        sc.Wcat = pk.Wcat; sc.WcatT = pk.WcatT; sc.colp = pk.colp; sc.sig = pk.sig; sc.rsig = pk.rsig;
        sc.Wp = pk.Wp; sc.WTp = pk.WTp;
    } else {
        ProfScope ps(PROF_PACK, s);
        pack_params_kernel<<<ew_grid(d.wslab), 256, 0, s>>>(pp, d.vc, in, d.ld_in, d.fp, sc.Wcat, sc.WcatT, sc.colp, sc.sig, sc.rsig,
                                                            BxOut{d.np ? sc.Wp : nullptr, d.wpslab, d.np, d.ld_in},
                                                            BxOut{d.np ? sc.WTp : nullptr, d.wtpslab, d.np, d.fp});
    }
    EAGCN_LAUNCH_CHECK();
    double fsum = 0.0;
    for (int k = 0; k < p->K; ++k) fsum += p->width[k];
    const double gemm_work = 2.0 * (double)b->T * (double)d.fin * fsum;

    BwdArgs ba;
    ba.meta = b->meta; ba.Tcap = b->T; ba.row_mol = b->row_mol; ba.row_m = b->row_m;
    ba.vc = d.vc; ba.structure = p->structure; ba.fp = d.fp;
    ba.nvirt = (dpad_row && p->structure == EAGCN_STRUCT_WEIGHTED) ? 1 : 0;
    ba.dxout = dxout; ba.ldo = d.ldo; ba.dpad = dpad_row; ba.dpad_views = dpad_views ? 1 : 0;
    if (rg) ba.rg = *rg; else memset(&ba.rg, 0, sizeof(ba.rg)); ba.Y = w->Y; ba.ldy = d.fp; ba.bn = w->bn;
    ba.colp = sc.colp; ba.dH = sc.dY; ba.slab = sc.slab; ba.slab_da = sc.slab_da;
    ba.do_drop = (p->training && p->dropout > 0.0f) ? 1 : 0;
    ba.thr = (uint32_t)std::min(4294967295.0, (double)p->dropout * 4294967296.0);
    ba.inv_keep = 1.0f / (1.0f - p->dropout);
    ba.seed = p->seed;
    ba.seed_dev = p->seed_dev;
    ba.zero = zero_after ? zero_after->d : nullptr;
    ba.nzero = zero_after ? zero_after->nd : 0;
    if (drain_in && drain_in->eacc) ba.drain = *drain_in; else memset(&ba.drain, 0, sizeof(ba.drain));
    const int rows = b->T + ba.nvirt;
    const int gxb = std::max(1, std::min(rows, d.gxb));
    bool fused_bn_apply = false, fused_w_apply = false, w_rows = false;
    // the LDS-staged aggregation (lagg.hip) runs this layer's transposed aggregation + edge gradients: decided ONCE, here, because the
    // BatchNorm backward below leaves its second pass to that kernel (the edge gradients must then leave through the shared accumulators)
    bool lagg_bwd = false;
    {
        static const bool edge_atomic0 = [] { const char* v = getenv("EAGCN_EDGE_SLABS"); return !(v && v[0] == '1'); }();
        bool general0 = false;
        for (int k = 0; k < p->K; ++k) general0 = general0 || pp.rel_vec[k] != nullptr;
        const bool wt0 = p->structure == EAGCN_STRUCT_WEIGHTED;
        const bool absorbs = lagg_fuses_bn() && (!bn_bwd_two_pass(wt0) || (wt0 && lagg_wfuse()));
        lagg_bwd = b->T > 0 && edge_atomic0 && !general0 && lagg_use(b, 1, absorbs);
    }
    const double M = (double)b->B * (double)b->N;
    {
        ProfScope ps(PROF_BN, s);
        const int ny = cdiv(d.fp, 1024);
        {
            const bool wt = p->structure == EAGCN_STRUCT_WEIGHTED, dr = ba.do_drop != 0;
            bool dg = ba.rg.dg != nullptr;
            const dim3 grid(gxb, ny);
            ba.cc = sc.cc;
            ba.store_dh = bn_bwd_two_pass(wt) ? 0 : 1;
            // Weighted_sum top layer whose dH is re-formed in lagg.hip's staging: that kernel wants the per-molecule read-out gradient as
            // ROWS ([T][ldo], K times narrower than dH) -- written first, so that this pass reads them with 16-byte loads as well
            // instead of gathering dg[molecule] element by element (fingerprint widths are not multiples of four: HIV 250)
            static const bool rows_first = [] { const char* v = getenv("EAGCN_W_ROWS_FIRST"); return !(v && v[0] == '0'); }();
            if (wt && dg && b->T > 0 && bn_bwd_two_pass(wt) && lagg_bwd && lagg_fuses_bn() && lagg_wfuse() && rows_first) {
                rc = launch_readout_bwd_rows(b, ba.rg, d.ldo, sc.dY, s);
                if (rc) return rc;
                w_rows = true;
                ba.dxout = sc.dY;
                dg = false;
            }
            launch_bn_bwd_pass(false, wt, dg, dr, grid, ba, s);
        }
        EAGCN_LAUNCH_CHECK();
        const double* gsum = nullptr;
        if (w->stats_hook && p->training) {
            // sync-BatchNorm: sum dH, sum dH xhat and the row count of this rank -> summed across the ranks by the caller's hook
            if (gxb > 64) bn_stats_sum_kernel<64><<<cdiv(d.fp, 4), 256, 0, s>>>(sc.slab, gxb, d.fp, M, sc.gsum, b->meta, 2, 0, ba.nvirt, b->B);
            else bn_stats_sum_kernel<16><<<cdiv(d.fp, 16), 256, 0, s>>>(sc.slab, gxb, d.fp, M, sc.gsum, b->meta, 2, 0, ba.nvirt, b->B);
            EAGCN_LAUNCH_CHECK();
            if (w->stats_hook(sc.gsum, 2 * d.fp + 1, stream, w->stats_user)) {
                set_error("eagcn_layer_backward: the sync-BatchNorm all-reduce hook failed");
                return EAGCN_ERR_HIP;
            }
            gsum = sc.gsum;
        }
        if (gxb > 64)
            bn_bwd_finalize_kernel<64><<<cdiv(d.fp, 4), 256, 0, s>>>(sc.slab, sc.slab_da, gxb, d.fp, M, p->training, w->bn,
                                                                       d.vc, gp, sc.cc, b->meta, ba.nvirt, b->B, ny, gxb, gsum, ba.zero, ba.nzero);
        else
            bn_bwd_finalize_kernel<16><<<cdiv(d.fp, 16), 256, 0, s>>>(sc.slab, sc.slab_da, gxb, d.fp, M, p->training, w->bn,
                                                                        d.vc, gp, sc.cc, b->meta, ba.nvirt, b->B, ny, gxb, gsum, ba.zero, ba.nzero);
        EAGCN_LAUNCH_CHECK();
        if (b->T > 0) {
            // second pass of the same kernel body: dH formed again, dY' written (sc.dY)
            const bool wt = p->structure == EAGCN_STRUCT_WEIGHTED, dg = ba.rg.dg != nullptr, dr = ba.do_drop != 0;
            if (bn_bwd_two_pass(wt) && wt && lagg_bwd && lagg_fuses_bn() && lagg_wfuse()) {
                fused_w_apply = true;                                           // the LDS-staged aggregation forms dH AND dY' in its staging
                if (dg && !w_rows) {                                                       // (per-molecule upstream gradient: its rows, once, K times narrower than dH)
                    rc = launch_readout_bwd_rows(b, ba.rg, d.ldo, sc.dY, s);
                    if (rc) return rc;
                }
            } else if (bn_bwd_two_pass(wt)) launch_bn_bwd_pass(true, wt, dg, dr, dim3(std::max(1, std::min(b->T, d.gxb)), ny), ba, s);
            else if (lagg_bwd && lagg_fuses_bn()) fused_bn_apply = true;      // the LDS-staged aggregation forms dY' from the stored dH while it stages its rows
            else bn_bwd_apply_kernel<<<ew_grid((size_t)b->T * d.fp / 4), 256, 0, s>>>(*b, d.fp, w->Y, d.fp, w->bn, sc.cc, sc.dY);
            EAGCN_LAUNCH_CHECK();
        }
    }
    int nsplit = 0, nedge = 0, xk_G = 0;
    bool bx_slabs = false;       // dWcat holds the k-chunk slabs of the plane GEMM (all of them written)
    int bx_per = 0;              // > 0: that product shared its launch with dX (workgroups per XCD: bx3.h bx3_used_splits)
    int bx_wide = 0;             // ... and which of the two plane-GEMM kernels wrote the slabs (the chunk policy counts ITS tiles)
    // side = stream for work that is off the dX critical path (edge gradients, dW product, gradient
    // unpacking); with no auxiliary stream everything stays in order on s
    hipStream_t side = w->aux_stream ? (hipStream_t)w->aux_stream : s;
    const bool forked = side != s;
    if (b->T > 0) {
        AggArgs a;
        a.bt = *b; a.vc = d.vc; a.src = sc.dY; a.lds = d.fp; a.dst = sc.dP; a.ldd = d.fp;
        a.sig = sc.sig; a.rsig = sc.rsig; a.rscale = w->rscale; a.stats = nullptr; a.nchunk = 1;
        // plane GEMMs (gemm_bx3.hip): dX = dP.Wcat^T (NT) and dW = X^T.dP (TN, k-chunk slabs summed by unpack_grads) in one
        // persistent launch; dP leaves the transposed aggregation as bf16 planes and is never written as fp32
        const size_t xstride = bx_plane_elems(b->T, d.ld_in), pstride = bx_plane_elems(b->T, d.fp);
        BxProb bx, bw;
        memset(&bx, 0, sizeof(bx));
        memset(&bw, 0, sizeof(bw));
        bool use_bx = false;
        if (d.np && !w->aux_stream) {
            bx.A = BxPlanes{sc.dPp, pstride, d.fp, b->T}; bx.B = BxPlanes{sc.Wp, d.wpslab, d.fp, d.ld_in}; bx.C = dx; bx.ldc = d.ld_in;
            bx.M = b->T; bx.N = d.ld_in; bx.K = d.fp; bx.M_dev = b->meta + EAGCN_META_T; bx.tn = 0; bx.splits = 1;
            bw.A = BxPlanes{w->x_planes ? w->x_planes : sc.xp, xstride, d.ld_in, b->T}; bw.B = BxPlanes{sc.dPp, pstride, d.fp, b->T};
            bw.C = sc.dWcat; bw.ldc = d.fp; bw.M = d.ld_in; bw.N = d.fp; bw.K = b->T; bw.K_dev = b->meta + EAGCN_META_T; bw.tn = 1;
            bw.splits = d.bx_splits; bw.slab = d.wslab;
            use_bx = bx3_ok(bw) && (!dx || bx3_ok(bx));
        }
        if (use_bx) {
            a.planes = BxOut{sc.dPp, pstride, d.np, b->T};
            if (!w->x_planes) {
                rc = launch_bx3_split(w->x, b->T, b->meta + EAGCN_META_T, d.ld_in, sc.xp, xstride, b->T, d.np, s);
                if (rc) return rc;
            }
        } else if (!w->x) {
            set_error("eagcn_layer_backward: the input exists as plane images only and the plane products do not apply");
            return EAGCN_ERR_ARG;
        }
        EdgeArgs e;
        e.bt = *b; e.vc = d.vc; e.dY = sc.dY; e.Y = w->Y; e.P = w->P; e.ld = d.fp; e.sig = sc.sig;
        e.rsig = sc.rsig; e.rscale = w->rscale; e.datt = sc.datt; e.atomic = 0;
        static const bool edge_atomic = [] { const char* v = getenv("EAGCN_EDGE_SLABS"); return !(v && v[0] == '1'); }();
        bool general_rel = false;                     // (code books: the reduction reads a view's whole histogram per channel)
        for (int k = 0; k < p->K; ++k) general_rel = general_rel || pp.rel_vec[k] != nullptr;
        if (edge_atomic && !general_rel) { e.datt = sc.eacc; e.atomic = 1; }
        static const bool colaunch = [] { const char* v = getenv("EAGCN_NO_COLAUNCH"); return !(v && v[0] == '1'); }();
        nedge = e.atomic ? -EDGE_COPIES : edge_grid_x(b);
        if (lagg_bwd && e.atomic) {                                              // transposed aggregation + edge gradients from the same LDS gathers
            if (fused_bn_apply || fused_w_apply) { a.bn_tab = w->bn; a.bn_cc = sc.cc; a.bn_fp = d.fp; }
            if (fused_w_apply) {
                a.src = (ba.rg.dg || w_rows) ? sc.dY : dxout; a.lds = d.ldo;
                a.w_aw = sc.colp + (size_t)CP_AVEW * d.fp;
                a.w_drop = ba.do_drop; a.w_thr = ba.thr; a.w_inv_keep = ba.inv_keep; a.w_seed = ba.seed; a.w_seed_dev = ba.seed_dev;
            }
            rc = launch_lagg_bwd(a, e, s);
            if (rc) return rc;
        } else if (!forked && colaunch) {
            rc = launch_agg_edge(a, e, s);                                       // one grid for both
            if (rc) return rc;
        } else {
            if (forked) { rc = stream_after(side, s); if (rc) return rc; }      // dY' is ready
            rc = launch_edge_grad(e, side);
            if (rc) return rc;
            rc = launch_agg(a, true, s);
            if (rc) return rc;
        }
        if (forked) { rc = stream_after(side, s); if (rc) return rc; }          // dP is ready
        nsplit = d.nsplit;
        GemmDesc gw{1, 0, d.ld_in, d.fp, b->T, w->x, d.ld_in, sc.dP, d.fp, sc.dWcat, d.fp, nsplit, d.wslab, gemm_work};
        gw.K_dev = b->meta + EAGCN_META_T;
        GemmDesc gx{0, 1, b->T, d.ld_in, d.fp, sc.dP, d.fp, sc.Wcat, d.fp, dx, d.ld_in, 1, 0, gemm_work};
        gx.M_dev = b->meta + EAGCN_META_T;
        static const bool pair = [] { const char* v = getenv("EAGCN_NO_PAIR"); return !(v && v[0] == '1'); }();
        static const bool use3 = [] { const char* v = getenv("EAGCN_NO_GEMM3"); return !(v && v[0] == '1'); }();
        DwScatter dsc;
        for (int k = 0; k < EAGCN_MAX_VIEWS; ++k) dsc.dW[k] = gp.dW[k];
        dsc.vc = d.vc;
        dsc.in = in;
        if (use_bx) {
            bx_wide = dx ? bx3_pick_wide(bx, &bw, b->t_hint) : bx3_pick_wide(bw, nullptr, b->t_hint);
            rc = dx ? launch_bx3(bx, &bw, d.np, s, 2.0 * gemm_work, PROF_GEMM_PAIR, bx_wide) : launch_bx3(bw, nullptr, d.np, s, gemm_work, PROF_GEMM, bx_wide);
            if (rc) return rc;
            nsplit = d.bx_splits;                         // partial slabs of dWcat: summed and scattered by unpack_grads below
            bx_slabs = true;
            bx_per = (dx && bx3_pair_policy()) ? std::max(1, bx3_grid() >> 3) : 0;
        } else if (use3 && !forked && dx && gemm3_layer(d.ld_in) && gemm3_ok(gw) && gemm3_ok(gx) && gemm3_xk_enabled()) {
            // wave-autonomous balanced kernel, XCD-local schedule: dX and dW in one launch, every XCD works on its own eighth of
            // the packed rows for BOTH products; dW leaves as one partial slab per XCD, summed by unpack_grads below
            rc = launch_gemm3_pair(gx, gw, nullptr, sc.gws, gemm3_workspace_bytes(), s, d.wslab);
            if (rc) return rc;
            nsplit = 1;
            xk_G = gemm3_grid();
        } else if (use3 && !forked && dx && gemm3_layer(d.ld_in) && gemm3_ok(gw) && gemm3_ok(gx)) {
            // ... tile-contiguous schedule: dW written straight into the per-view gradients
            rc = launch_gemm3_pair(gx, gw, &dsc, sc.gws, gemm3_workspace_bytes(), s);
            if (rc) return rc;
            nsplit = 0;                                   // no partial slabs: unpack_grads only reduces the edge partials
        } else if (dx && !forked && colaunch && pair) {
            rc = launch_gemm_pair(gx, gw, s);                                    // dX and dW share one grid
            if (rc) return rc;
        } else {
            rc = launch_gemm(gw, side);
            if (rc) return rc;
            if (dx) { rc = launch_gemm(gx, s); if (rc) return rc; }
        }
    }
    if (b->T == 0) {
        // a batch without a single bond has no packed row: no product ran, the weight gradients are exactly zero (the edge
        // partials below are summed over zero workgroups; the BatchNorm gradients came from the row-less reduction above)
        for (int k = 0; k < p->K; ++k)
            { int rcz = zero_fill(gp.dW[k], (size_t)d.fin * p->width[k] * sizeof(float), side); if (rcz) return rcz; }
    }
    {
        const int wblocks = nsplit > 0 ? cdiv((int)d.wslab, 256) : 0;     // (one thread per element: four lanes per four elements, unpack_grads)
        double* edge_src = nedge == -EDGE_COPIES ? sc.eacc : sc.datt;
        const int edge_drain = edge_src == sc.eacc ? 1 : 0;       // shared accumulators: zeroed again by the threads that read them
        static const bool defer_env = [] { const char* v = getenv("EAGCN_NO_EDGE_DEFER"); return !(v && v[0] == '1'); }();
        if (defer_env && drain_out && edge_drain && wblocks == 0 && !forked && b->T > 0) {
            // nothing but the edge reduction is left for this layer: the next layer's first backward kernel does it (one launch less)
            drain_out->eacc = sc.eacc; drain_out->K = p->K; drain_out->rsig = sc.rsig;
            for (int k = 0; k < EAGCN_MAX_VIEWS; ++k) {
                drain_out->datt_w[k] = gp.datt_w[k]; drain_out->dself_r[k] = gp.dself_r[k]; drain_out->channels[k] = pp.channels[k];
            }
            return EAGCN_OK;
        }
        ProfScope psu(PROF_PACK, side);
        unpack_grads_kernel<<<wblocks + cdiv(p->K * EDGE_SLAB, 16), 256, 0, side>>>(gp, pp, d.vc, in, d.ld_in, d.fp, sc.dWcat,
                                                                                    bx_slabs ? -nsplit : nsplit, d.wslab, edge_src, nedge, sc.rsig,
                                                                                    wblocks, b->meta, xk_G, edge_drain, bx_per, bx_wide);
        EAGCN_LAUNCH_CHECK();
    }
    // join: the caller reuses the scratch block (dY', dP, partial slabs) for the next layer
    if (forked) { rc = stream_after(s, side); if (rc) return rc; }
    return EAGCN_OK;
}

extern "C" int eagcn_attention_dense(const eagcn_batch* b, const eagcn_layer_params* p, float* out, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    EAGCN_CHECK_ARG(b && p && out, "eagcn_attention_dense: null argument");
    EAGCN_CHECK_ARG(p->K == b->K, "eagcn_attention_dense: view count mismatch");
    EAGCN_CHECK_ARG(b->code, "eagcn_attention_dense: batch has no code map");
    ParamPtrs pp;
    memset(&pp, 0, sizeof(pp));
    for (int k = 0; k < p->K; ++k) {
        EAGCN_CHECK_ARG(p->att_w[k], "eagcn_attention_dense: view %d has no attention weight", k);
        pp.att_w[k] = p->att_w[k];
        pp.channels[k] = b->channels[k];
        pp.rel_vec[k] = b->rel_vec[k]; pp.rel_c[k] = b->rel_c[k];
    }
    const size_t total = (size_t)b->K * b->B * b->N * b->N;
    attention_dense_kernel<<<ew_grid(total), 256, 0, s>>>(*b, pp, out);
    EAGCN_LAUNCH_CHECK();
    return EAGCN_OK;
}
