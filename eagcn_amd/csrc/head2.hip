// The model head (reference models.py:112-120): Graph_BN -> den1 -> bn_den1 -> relu -> dropout -> den2
// (= graph_representation) -> bn_den2 -> relu -> den3, and its backward, as FOUR launches per direction.
//
// A BatchNorm1d over a [B, F] matrix needs column statistics over the whole batch before a single output can be
// formed, so a chain  BN -> Dense -> BN -> Dense ...  looks like one launch per BatchNorm plus one per product (round 1:
// 13 launches of 5-15 us for 0.4 % of the step's flops).  Here every BatchNorm is split in two halves that ride in the
// neighbouring products:
//   * its STATISTICS (sum x, sum x^2 per column; backward: sum dy, sum dy*xhat) are accumulated in fp64 atomics by the
//     epilogue of the kernel that PRODUCES x (read-out, den1, den2; backward: the kernel that produces dy);
//   * its NORMALISATION (+ relu, dropout; backward: the affine dx = a*dy + b*x + c) is applied by the kernel that
//     CONSUMES it, while it loads its operands (per-column tables in LDS, built from the statistics by every workgroup).
// The products themselves are tiny (B x 700 x 256 ...): one WAVE per 16x16 output tile, operands straight from global
// memory into the MFMA operand registers (v_mfma_f32_16x16x4_f32, exact fp32); one WORKGROUP per 16x64 output tile, its
// four waves split K and keep two k-steps (ten vector loads) in flight each (the products are latency chains, not flops).
// Every backward launch carries two products: d(input) of a dense layer (NT form) and its weight gradient (TN form).
#include <stdlib.h>

#include <algorithm>

#include "common.h"
#include "kernels.h"

namespace eagcn {

enum { HT_SC = 0, HT_SH, HT_MU, HT_INV };    // rows of a saved BatchNorm table [4][F]

__device__ __forceinline__ uint64_t head_seed(const HeadDrop& d) { return d.seed_dev ? *d.seed_dev : d.seed; }

// ---- 16x64 output tile of one WORKGROUP: acc[j] = sum_k A(row, k) B_j(k, col) ----------------------------------------------
// MFMA step s of a 16-wide k-step multiplies the actual k = k0 + 4q + s in slot q for both operands.  The operands are
// given as functors with a LOAD half (raw registers, clamped addresses, no predicate) and a TRANSFORM half (BatchNorm
// affine / relu / dropout / zero beyond the extents): the products are chains of dependent memory round trips, not of
// MFMAs, so all loads of STEPS k-steps are issued back to back before the first value is touched (a conditional or a
// select right behind a load makes the compiler wait -- or branch -- per load).  Two forms of the B side:
//   QUAD: fb.load(k) = B[k][n0 + 4 li .. + 3], one k, four adjacent columns: component j feeds column tile j (tile j
//         holds the columns 4c + j; 16 lanes read 256 contiguous bytes of a [K][N] operand);
//   NT:   fb.load(j, kq) = the four values k = kq .. kq+3 of column 16 j + li (an operand stored [N][K]).
// The waves of the workgroup split the k-steps; the partial tiles are summed through LDS and returned to wave 0.
__device__ __forceinline__ void keep(f32x4& v) { asm volatile("" : "+v"(v)); }   // the load feeding v is not sunk / predicated

// The reduction runs over k0 <= k < k1 (k0 a multiple of 16; the functors zero whatever lies beyond the operands' extents).
// A workgroup is HT_NW = 8 waves (round 6; four before): wave w takes the k-steps [w per, (w + 1) per), per = ceil(steps / 8) -- the
// 700-wide first product is ONE batch of loads per wave instead of three, the 256-long reductions of the backward one instead
// of two.  begin() requests a wave's first batch, finish() transforms, multiplies, walks the remaining batches and adds the
// waves' partial tiles up in wave 0 (a fixed tree): the kernels put their BatchNorm table build BETWEEN the two, so that the
// table's own loads (the fp64 sums) and the first operand batch are one memory round trip, not two.
// NW = 4 where a launch has more tiles than the chip has CUs (eight waves of these register counts are one workgroup per CU: dense 1's
// backward, 352 tiles at configs[1], took two rounds and 19.6 instead of 14.7 us); the stages the fused middle launch shares
// with the separate launches (dense 3) are always 8 wide, so that both forms add in the same order (HeadFwd.nw / HeadBwd.nw).
constexpr int HT_NW = 8;
constexpr int HT_NT = 64 * HT_NW;            // threads per workgroup (the wide form)
constexpr int HT_RED = HT_NW * 1024;         // floats of the partial-tile buffer (4 tiles x 64 lanes x 4 per wave)
template <bool QUAD, int STEPS, int NW, class FA, class FB>
struct HeadTile {
    typename FA::Raw ra[STEPS];
    typename FB::Raw rb[STEPS][4];
    int k0, sb, se;
    __device__ __forceinline__ void issue(int st, const FA& fa, const FB& fb) {
        const int q = (threadIdx.x & 63) >> 4;
#pragma unroll
        for (int u = 0; u < STEPS; ++u) {
            const int kq = k0 + (st + u) * 16 + 4 * q;     // (steps beyond the wave's range: clamped addresses, never multiplied)
            ra[u] = fa.load(kq);
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                if constexpr (QUAD) rb[u][t] = fb.load(kq + t);   // t = MFMA step s
                else rb[u][t] = fb.load(t, kq);                   // t = column tile j
            }
        }
    }
    __device__ __forceinline__ void begin(int k0_, int k1, const FA& fa, const FB& fb) {
        const int wave = threadIdx.x >> 6;
        const int nsteps = max(0, (k1 - k0_ + 15) >> 4);
        const int per = (nsteps + NW - 1) / NW;
        k0 = k0_;
        sb = wave * per;
        se = min(nsteps, sb + per);
        if (sb < se) issue(sb, fa, fb);
    }
    __device__ __forceinline__ void finish(const FA& fa, const FB& fb, f32x4* red, f32x4 (&acc)[4]) {
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, q = lane >> 4;
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        for (int st = sb; st < se; st += STEPS) {
            if (st != sb) issue(st, fa, fb);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < STEPS; ++u) {
                if (st + u < se) {                            // (wave-uniform)
                    const int kq = k0 + (st + u) * 16 + 4 * q;
                    const f32x4 a = fa.xf(ra[u], kq);
                    f32x4 b[4];
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        if constexpr (QUAD) b[t] = fb.xf(rb[u][t], kq + t);
                        else b[t] = fb.xf(rb[u][t], t, kq);
                    }
#pragma unroll
                    for (int s = 0; s < 4; ++s)
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s], QUAD ? b[s][j] : b[j][s], acc[j], 0, 0, 0);
                }
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) red[(j * NW + wave) * 64 + lane] = acc[j];
        __syncthreads();
        if (wave == 0) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const f32x4* r = red + (j * NW) * 64 + lane;
                if constexpr (NW == 8) acc[j] = ((r[0] + r[64]) + (r[128] + r[192])) + ((r[256] + r[320]) + (r[384] + r[448]));
                else acc[j] = (r[0] + r[64]) + (r[128] + r[192]);
            }
        }
    }
};
struct NoMid { __device__ __forceinline__ void operator()() const {} };

// raw load of p[i .. i+3] at a clamped (always valid) index; VEC: i and n multiples of 4, row 16-byte aligned
template <bool VEC>
__device__ __forceinline__ f32x4 ld4raw(const float* __restrict__ p, int i, int n) {
    f32x4 v;
    if constexpr (VEC) {
        v = *reinterpret_cast<const f32x4*>(p + max(min(i, n - 4), 0));
    } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = p[max(min(i + e, n - 1), 0)];
    }

    return v;
}
__device__ __forceinline__ f32x4 zero_beyond(f32x4 v, int i, int n) {
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = i + e < n ? v[e] : 0.0f;
    return v;
}

// ---- forward stage: y = act(bn(x)) . W, statistics of y (HeadFwd, kernels.h) ------------------------------------------------
template <bool VEC, bool DROP>
struct HfA {                     // A side: x row, BatchNorm affine from the LDS table, relu, dropout
    typedef f32x4 Raw;
    const float* xr; const float* tab; int K, Kp, row; float lo; uint64_t seed; uint32_t thr; float inv_keep;
    __device__ __forceinline__ Raw load(int kq) const { return ld4raw<VEC>(xr, kq, K); }
    __device__ __forceinline__ f32x4 xf(Raw v, int kq) const {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int k = min(kq + e, K - 1);
            float h = fmaxf(v[e] * tab[k] + tab[Kp + k], lo);
            if constexpr (DROP) h *= drop_scale(seed, (uint64_t)row * K + k, thr, inv_keep);
            v[e] = kq + e < K ? h : 0.0f;
        }
        return v;
    }
};
template <bool VEC>
struct HfB {                     // B side (QUAD): W[k][col0 .. col0+3]
    typedef f32x4 Raw;
    const float* W; int K, N, col0;
    __device__ __forceinline__ Raw load(int k) const { return ld4raw<VEC>(W + (size_t)min(k, K - 1) * N, col0, N); }
    __device__ __forceinline__ f32x4 xf(Raw v, int k) const { return k < K ? zero_beyond(v, col0, N) : (f32x4){0.f, 0.f, 0.f, 0.f}; }
};

// BatchNorm table of a forward stage, built by every workgroup that has tiles of it: tab[0..Kp) scale, tab[Kp..2Kp) shift.
// `writer` (one workgroup per stage) also updates the running statistics and saves the [4][K] table for the backward.
// `full`: mu and 1 / sigma follow in tab[2Kp..4Kp) (the rows of the saved table, for a backward tile of the same workgroup).
__device__ __forceinline__ void hf_table(const HeadFwd& a, float* tab, bool writer, bool full = false) {
    const int Kp = (a.K + 3) & ~3;
    const double Bn = (a.cnt_in && a.training) ? *a.cnt_in : (double)a.B;      // rows of the BatchNorm (all ranks with sync-BatchNorm)
    for (int k = threadIdx.x; k < a.K; k += (int)blockDim.x) {
        float mu, inv;
        if (a.training) {
            double v1[HEAD_COPIES], v2[HEAD_COPIES];                  // all replicas' loads in flight together
#pragma unroll
            for (int c = 0; c < HEAD_COPIES; ++c) {
                const size_t o = (size_t)(c < a.st_copies ? c : 0) * a.st_stride + 2 * k;
                v1[c] = a.st_in[o];
                v2[c] = a.st_in[o + 1];
            }
            double t1 = v1[0], t2 = v2[0];
#pragma unroll
            for (int c = 1; c < HEAD_COPIES; ++c) {
                t1 += c < a.st_copies ? v1[c] : 0.0;
                t2 += c < a.st_copies ? v2[c] : 0.0;
            }
            const double mean = t1 / Bn;
            double var = t2 / Bn - mean * mean;
            var = var > 0.0 ? var : 0.0;
            mu = (float)mean;
            inv = (float)(1.0 / sqrt(var + (double)a.eps));
            if (writer) {
                const double unbiased = var * (Bn / (Bn - 1.0));
                a.run_mean[k] = (float)((1.0 - a.momentum) * (double)a.run_mean[k] + a.momentum * mean);
                a.run_var[k] = (float)((1.0 - a.momentum) * (double)a.run_var[k] + a.momentum * unbiased);
            }
        } else {
            mu = a.run_mean[k];
            inv = 1.0f / sqrtf(a.run_var[k] + a.eps);
        }
        const float sc = a.gamma[k] * inv, sh = a.beta[k] - mu * sc;
        tab[k] = sc;
        tab[Kp + k] = sh;
        if (full) { tab[2 * Kp + k] = mu; tab[3 * Kp + k] = inv; }
        if (writer) {
            a.bn[HT_SC * a.K + k] = sc; a.bn[HT_SH * a.K + k] = sh; a.bn[HT_MU * a.K + k] = mu; a.bn[HT_INV * a.K + k] = inv;
        }
    }
}

// one 16 x 64 output tile of a forward stage: the product (all four waves; the sum arrives in wave 0's `acc`) ...
// (`mid` runs between the request of the first operand batch and its use: the kernels' table build + barrier)
// (STEPS: k-steps of a wave requested together -- six cover the 700-wide first product in one batch; three where the dropout hash or
//  scalar loads need the registers)
template <bool VEC, bool DROP, int STEPS = 3, int NW = HT_NW, class Mid = NoMid>
__device__ __forceinline__ void hf_tile(const HeadFwd& a, int tile, const float* tab, f32x4* red, f32x4 (&acc)[4], Mid mid = Mid()) {
    const int Kp = (a.K + 3) & ~3;
    const int li = threadIdx.x & 15;
    const int ncb = (a.N + 63) >> 6;
    const int rb = tile / ncb, cb = tile - rb * ncb;
    const int row = min(rb * 16 + li, a.B - 1), col0 = cb * 64 + 4 * li;
    const HfA<VEC, DROP> fa{a.x + (size_t)row * a.K, tab, a.K, Kp, row, a.relu ? 0.0f : -INFINITY,
                            DROP ? head_seed(a.drop) : 0, a.drop.thr, a.drop.inv_keep};
    const HfB<VEC> fb{a.W, a.K, a.N, col0};
    HeadTile<true, STEPS, NW, HfA<VEC, DROP>, HfB<VEC>> t;
    t.begin(0, a.K, fa, fb);
    mid();
    t.finish(fa, fb, red, acc);
}
// ... and its epilogue (wave 0 only): y (+ its copy y2), column sums of y into one replica of st_out
__device__ __forceinline__ void hf_store(const HeadFwd& a, int tile, const f32x4 (&acc)[4]) {
    const int lane = threadIdx.x & 63, li = lane & 15, q = lane >> 4;
    const int ncb = (a.N + 63) >> 6;
    const int rb = tile / ncb, cb = tile - rb * ncb;
    const int col0 = cb * 64 + 4 * li;
    // D layout of tile j: column 4 li + j, rows 4q + r
    double s1[4] = {0.0, 0.0, 0.0, 0.0}, s2[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int orow = rb * 16 + 4 * q + r;
        if (orow < a.B) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (col0 + j < a.N) {
                    const float v = acc[j][r];
                    a.y[(size_t)orow * a.N + col0 + j] = v;
                    if (a.y2) a.y2[(size_t)orow * a.N + col0 + j] = v;
                    s1[j] += (double)v;
                    s2[j] += (double)v * (double)v;
                }
        }
    }
    if (a.st_out) {
        double* so = a.st_out + (size_t)(rb % a.st_copies) * a.st_stride;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            double t1 = s1[j], t2 = s2[j];
            t1 += __shfl_xor(t1, 16); t2 += __shfl_xor(t2, 16);
            t1 += __shfl_xor(t1, 32); t2 += __shfl_xor(t2, 32);
            if (q == 0 && col0 + j < a.N) {
                atomicAdd(&so[2 * (col0 + j)], t1);
                atomicAdd(&so[2 * (col0 + j) + 1], t2);
            }
        }
    }
}

template <bool VEC, bool DROP, int NW>
__global__ __launch_bounds__(64 * NW) void head_fwd_kernel(HeadFwd a) {
    extern __shared__ __attribute__((aligned(16))) float tab[];          // [2][Kp]: scale, shift
    const int Kp = (a.K + 3) & ~3;
    // (relaxed: the word only PLACES the side stream's work, nothing it guards is read through it)
    if (a.signal && blockIdx.x == 0 && threadIdx.x == 0) __hip_atomic_fetch_add(a.signal, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (a.lab) {                                                         // this workgroup's share of the batch's labelled entries (BCE)
        const int per = (a.nlab + gridDim.x - 1) / gridDim.x;
        const int i = blockIdx.x * per + threadIdx.x;
        int c = 0;
        for (int j = i; j < min(a.nlab, ((int)blockIdx.x + 1) * per); j += 64 * NW) {
            const float v = a.lab[j];
            c += (v == 1.0f || v == 0.0f) ? 1 : 0;
        }
        c = wave_sum(c);
        if ((threadIdx.x & 63) == 0 && c) atomicAdd(a.lab_cnt, (unsigned)c);
    }
    f32x4* red = reinterpret_cast<f32x4*>(tab + 2 * Kp);
    f32x4 acc[4];
    hf_tile<VEC, DROP, (VEC && !DROP && NW == 8) ? 6 : 3, NW>(a, blockIdx.x, tab, red, acc, [&]() {      // one 16 x 64 tile per workgroup
        hf_table(a, tab, blockIdx.x == 0);
        __syncthreads();
    });
    if (threadIdx.x < 64) hf_store(a, blockIdx.x, acc);
}

// ---- backward stage of the dense layer  y = a . W,  a = act(bn_p(x)) -----------------------------------------------------
//   dy_eff = the gradient that reaches y: either given plainly (dense 3: d out) or through the BatchNorm that follows y:
//            dy_eff = al * dy + be * y + ga (+ extra), coefficients from that BatchNorm's backward sums (this kernel's
//            prologue; it also writes that BatchNorm's d gamma / d beta)
//   (a) da = dy_eff . W^T ;  dyp = da * dropmask * [bn_p(x) > 0]  -> written, with sum dyp, sum dyp * xhat_p (atomics)
//   (b) dW = a^T . dy_eff
struct Raw3 { f32x4 d, y, e; };
template <bool VEC>
struct HbDy {                    // dy_eff[b][n4 .. n4+3]: the plain case reads dy in place of y with the coefficients
    typedef Raw3 Raw;            // (1, 0, 0), a missing `extra` reads dy with weight 0 -- no branch between the loads
    const float *dy, *yp, *ep; const float* tab; int B, N, Np; float ew;
    __device__ __forceinline__ Raw ld(int b, int n4) const {
        const size_t ro = (size_t)min(b, B - 1) * N;
        Raw r;
        r.d = ld4raw<VEC>(dy + ro, n4, N); r.y = ld4raw<VEC>(yp + ro, n4, N); r.e = ld4raw<VEC>(ep + ro, n4, N);
        return r;
    }
    __device__ __forceinline__ f32x4 tr(const Raw& r, int b, int n4) const {
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int n = min(n4 + e, N - 1);
            const float v = tab[n] * r.d[e] + tab[Np + n] * r.y[e] + tab[2 * Np + n] + ew * r.e[e];
            o[e] = (n4 + e < N && b < B) ? v : 0.0f;
        }
        return o;
    }
};
template <bool VEC>
struct HbA_a {                   // (a) A side: dy_eff[row][nq .. nq+3]
    typedef Raw3 Raw;
    HbDy<VEC> g; int row;
    __device__ __forceinline__ Raw load(int nq) const { return g.ld(row, nq); }
    __device__ __forceinline__ f32x4 xf(const Raw& r, int nq) const { return g.tr(r, row, nq); }
};
template <bool VEC>
struct HbB_a {                   // (a) B side (NT): W[k_j][nq .. nq+3], k_j = kb*64 + 16 j + li
    typedef f32x4 Raw;
    const float* W; int K, N, k0;
    __device__ __forceinline__ Raw load(int j, int nq) const { return ld4raw<VEC>(W + (size_t)min(k0 + 16 * j, K - 1) * N, nq, N); }
    __device__ __forceinline__ f32x4 xf(Raw v, int j, int nq) const { return zero_beyond(v, nq, N); }
};
template <bool DROP>
struct HbA_b {                   // (b) A side: a[b][k] = drop(relu(bn_p(x[b][k]))) for b = bq .. bq+3
    typedef f32x4 Raw;
    const float* x; int B, K, k; float sc, sh, lo; uint64_t seed; uint32_t thr; float inv_keep;
    __device__ __forceinline__ Raw load(int bq) const {
        f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = x[(size_t)min(bq + e, B - 1) * K + k];
        keep(v);
        return v;
    }
    __device__ __forceinline__ f32x4 xf(Raw v, int bq) const {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float h = fmaxf(v[e] * sc + sh, lo);
            if constexpr (DROP) h *= drop_scale(seed, (uint64_t)(bq + e) * K + k, thr, inv_keep);
            v[e] = bq + e < B ? h : 0.0f;
        }
        return v;
    }
};
template <bool VEC>
struct HbB_b {                   // (b) B side (QUAD): dy_eff[b][n0 .. n0+3]
    typedef Raw3 Raw;
    HbDy<VEC> g; int n0;
    __device__ __forceinline__ Raw load(int b) const { return g.ld(b, n0); }
    __device__ __forceinline__ f32x4 xf(const Raw& r, int b) const { return g.tr(r, b, n0); }
};

// coefficient table of a backward stage: tab[0..Np) al, [Np..2Np) be, [2Np..3Np) ga of dy_eff; `writer` also stores d gamma / d beta
__device__ __forceinline__ void hb_table(const HeadBwd& a, float* tab, bool writer) {
    const int Np = (a.N + 3) & ~3;
    const double Bn = (a.cnt_y && a.training) ? *a.cnt_y : (double)a.B;
    for (int n = threadIdx.x; n < a.N; n += (int)blockDim.x) {
        float al = 1.0f, be = 0.0f, ga = 0.0f;
        if (a.bny) {
            const float sc = a.bny[HT_SC * a.N + n], mu = a.bny[HT_MU * a.N + n], inv = a.bny[HT_INV * a.N + n];
            const double s1 = a.sb_y[2 * n], s2 = a.sb_y[2 * n + 1];
            const float c1 = a.training ? (float)(s1 / Bn) : 0.0f, c2 = a.training ? (float)(s2 / Bn) : 0.0f;
            // sc * (dy - c1 - (y - mu) * inv * c2)
            al = sc;
            be = -sc * inv * c2;
            ga = sc * (inv * c2 * mu - c1);
            if (writer) { a.dgamma_y[n] = (float)(s2 * (double)a.gscale); a.dbeta_y[n] = (float)(s1 * (double)a.gscale); }
        }
        tab[n] = al; tab[Np + n] = be; tab[2 * Np + n] = ga;
    }
}
__device__ __forceinline__ int hb_tiles_a(const HeadBwd& a) { return ((a.B + 15) >> 4) * ((a.K + 63) >> 6); }
__device__ __forceinline__ int hb_tiles_b(const HeadBwd& a) { return ((a.K + 15) >> 4) * ((a.N + 63) >> 6) * a.ks; }

// one tile of a backward stage: tile < hb_tiles_a: 16 x 64 of d(input) with its epilogue, else 16 x 64 of dW (chunk-major)
// bnp / bnp_ld: the [4][bnp_ld] table of the BatchNorm in front of the dense layer (a.bnp with stride K, or a copy in LDS)
template <bool VEC, bool DROP, int STEPS = (VEC ? 2 : 1), int NW = HT_NW, class Mid = NoMid>
__device__ __forceinline__ void hb_tile(const HeadBwd& a, int tile, const float* tab, f32x4* red, const float* bnp, int bnp_ld, Mid mid = Mid()) {
    const int Np = (a.N + 3) & ~3;
    const int lane = threadIdx.x & 63, li = lane & 15, q = lane >> 4;
    const int nkb64 = (a.K + 63) >> 6, nnb64 = (a.N + 63) >> 6, nrb = (a.B + 15) >> 4;
    const int tiles_a = nrb * nkb64;
    const uint64_t seed = DROP ? head_seed(a.drop) : 0;
    const HbDy<VEC> dyf{a.dy, a.bny ? a.y : a.dy, a.extra ? a.extra : a.dy, tab, a.B, a.N, Np, a.extra ? 1.0f : 0.0f};
    f32x4 acc[4];
    if (tile < tiles_a) {
        // (a) da[b][k] = sum_n dy_eff[b][n] W[k][n]   (both operands contiguous along the reduction index n: NT form)
        const int rb = tile / nkb64, kb = tile - rb * nkb64;
        const HbA_a<VEC> fa{dyf, min(rb * 16 + li, a.B - 1)};
        const HbB_a<VEC> fb{a.W, a.K, a.N, kb * 64 + li};
        HeadTile<false, STEPS, NW, HbA_a<VEC>, HbB_a<VEC>> t;
        t.begin(0, a.N, fa, fb);
        mid();
        t.finish(fa, fb, red, acc);
        if (threadIdx.x >= 64) return;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int k = kb * 64 + 16 * j + li;             // tile j: column li <-> input feature k
            double s1 = 0.0, s2 = 0.0;
            if (k < a.K) {
                const float sc = bnp[HT_SC * bnp_ld + k], sh = bnp[HT_SH * bnp_ld + k];
                const float mu = bnp[HT_MU * bnp_ld + k], inv = bnp[HT_INV * bnp_ld + k];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int b = rb * 16 + 4 * q + r;
                    if (b < a.B) {
                        const float xv = a.x[(size_t)b * a.K + k];
                        float d = acc[j][r];
                        if constexpr (DROP) d *= drop_scale(seed, (uint64_t)b * a.K + k, a.drop.thr, a.drop.inv_keep);
                        if (a.relu_p && !(xv * sc + sh > 0.0f)) d = 0.0f;
                        a.dyp[(size_t)b * a.K + k] = d;
                        s1 += (double)d;
                        s2 += (double)(d * ((xv - mu) * inv));
                    }
                }
            }
            s1 += __shfl_xor(s1, 16); s2 += __shfl_xor(s2, 16);
            s1 += __shfl_xor(s1, 32); s2 += __shfl_xor(s2, 32);
            if (q == 0 && k < a.K) {
                atomicAdd(&a.sb_p[2 * k], s1);
                atomicAdd(&a.sb_p[2 * k + 1], s2);
            }
        }
    } else {
        // (b) dW[k][n] = sum_b a[b][k] dy_eff[b][n],  a = drop(relu(bn_p(x)));  tile: 16 input features x 64 output features
        int t = tile - tiles_a;
        const int tiles_b = ((a.K + 15) >> 4) * nnb64;
        const int chunk = t / tiles_b;                    // row chunk of the reduction (HeadBwd.ks)
        t -= chunk * tiles_b;
        const int rows_per = ((nrb + a.ks - 1) / a.ks) * 16;
        const int r0 = chunk * rows_per, r1 = min(a.B, r0 + rows_per);
        float* __restrict__ dW = a.ks > 1 ? a.dW_part + (size_t)chunk * a.K * a.N : a.dW;
        const int kb = t / nnb64, nb = t - kb * nnb64;
        const int k = min(kb * 16 + li, a.K - 1), n0 = nb * 64 + 4 * li;
        const HbA_b<DROP> fa{a.x, a.B, a.K, k, bnp[HT_SC * bnp_ld + k], bnp[HT_SH * bnp_ld + k], a.relu_p ? 0.0f : -INFINITY,
                             seed, a.drop.thr, a.drop.inv_keep};
        const HbB_b<VEC> fb{dyf, n0};
        HeadTile<true, STEPS, NW, HbA_b<DROP>, HbB_b<VEC>> ht;
        ht.begin(r0, r1, fa, fb);
        mid();
        ht.finish(fa, fb, red, acc);
        if (threadIdx.x >= 64) return;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int ok = kb * 16 + 4 * q + r;
            if (ok < a.K) {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (n0 + j < a.N) dW[(size_t)ok * a.N + n0 + j] = acc[j][r];
            }
        }
    }
}

// (two waves per SIMD: with NW = 4 two workgroups per CU, the whole of a 352-tile launch resident at once)
template <bool VEC, bool DROP, int NW>
__global__ __launch_bounds__(64 * NW) __attribute__((amdgpu_waves_per_eu(2))) void head_bwd_kernel(HeadBwd a) {
    extern __shared__ __attribute__((aligned(16))) float tab[];          // [3][Np]: al, be, ga of dy_eff
    const int Np = (a.N + 3) & ~3;
    hb_tile<VEC, DROP, (VEC ? 2 : 1), NW>(a, blockIdx.x, tab, reinterpret_cast<f32x4*>(tab + 3 * Np), a.bnp, a.K, [&]() {     // one 16 x 64 tile per workgroup
        hb_table(a, tab, blockIdx.x == 0);
        __syncthreads();
    });
}

// multi-round launches (B >= 1024: 1 400 tiles of dense 1's backward on 512 resident workgroups): ONE k-step in flight per wave
// instead of two -> 150 instead of 250 registers -> three workgroups per CU instead of two (the 16-byte-load form only: the scalar form spills)
template <bool DROP>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3))) void head_bwd_light_kernel(HeadBwd a) {
    extern __shared__ __attribute__((aligned(16))) float tab[];
    const int Np = (a.N + 3) & ~3;
    hb_tile<true, DROP, 1, 4>(a, blockIdx.x, tab, reinterpret_cast<f32x4*>(tab + 3 * Np), a.bnp, a.K, [&]() {
        hb_table(a, tab, blockIdx.x == 0);
        __syncthreads();
    });
}

// Graph_BN backward (no product in front of it): dg = al * dgn + be * g + ga, d gamma / d beta; workgroup `wg` of `nwg`
__device__ __forceinline__ void hg_body(const HeadGbn& a, int wg, int nwg) {
    const size_t total = (size_t)a.B * a.F;
    const double Bn = (a.cnt && a.training) ? *a.cnt : (double)a.B;
    for (size_t e = (size_t)wg * 256 + threadIdx.x; e < total; e += (size_t)nwg * 256) {
        const int f = (int)(e % a.F);
        const float sc = a.bn[HT_SC * a.F + f], mu = a.bn[HT_MU * a.F + f], inv = a.bn[HT_INV * a.F + f];
        const double s1 = a.sb[2 * f], s2 = a.sb[2 * f + 1];
        const float c1 = a.training ? (float)(s1 / Bn) : 0.0f, c2 = a.training ? (float)(s2 / Bn) : 0.0f;
        a.dg[e] = sc * (a.dgn[e] - c1 - (a.g[e] - mu) * inv * c2);
        if (e < (size_t)a.F) { a.dgamma[f] = (float)(s2 * (double)a.gscale); a.dbeta[f] = (float)(s1 * (double)a.gscale); }
    }
    // weight gradients of the dense layers that left as row-chunk partials (HeadBwd.ks): summed in chunk order
    for (int j = 0; j < a.nsum; ++j) {
        const HeadDwSum q = a.sum[j];
        for (int i = wg * 256 + threadIdx.x; i < q.n; i += nwg * 256) {
            float v[16];
#pragma unroll
            for (int c = 0; c < 16; ++c) v[c] = q.part[(size_t)min(c, q.ks - 1) * q.n + i];
            float t = v[0];
#pragma unroll
            for (int c = 1; c < 16; ++c) t += c < q.ks ? v[c] : 0.0f;
            q.dst[i] = t;
        }
    }
}
__global__ __launch_bounds__(256) void head_gbn_bwd_kernel(HeadGbn a) { hg_body(a, blockIdx.x, gridDim.x); }

// ---- the middle of a TRAINING step's head as ONE launch: last forward stage, loss (train.py:321-331), first backward product --
// A 16-row block of the logits is ONE tile when nclass <= 64, the loss terms and d out of those rows need nothing but the rows
// (and the batch's count of labelled entries, which the launch in front -- dense 2 -- takes on the way: HeadFwd.lab), and
// d a2 = d out . W3^T needs nothing but those rows of d out: out = relu(bn2(h2)) . W3, the loss and dense 3's d(input) product run
// in the workgroup that owns the row block, with no launch boundary and no trip through memory between them (they were three
// launches of 6-10 us: three dependent chains of a table build, a load batch and an epilogue).  Same tile bodies and the same
// arithmetic as the separate launches.  dense 3's WEIGHT gradient (a reduction over all rows) rides in dense 2's backward launch.
// The loss value is the sum of the row blocks' partials (fp64 atomic on a word of the head's cleared sums block); the workgroup
// that finishes last publishes it.
//
// Measured and dropped (round 6): the WHOLE head -- three forward stages, loss, three backward stages, Graph_BN backward -- as one
// persistent launch with device-scope arrive / spin barriers between the phases.  Bit-identical, and SLOWER: 0.507 ms per
// configs[1] step against 0.400 with eight launches (0.467 with 128 workgroups, 0.66 with 16): a dependent launch boundary costs
// 1.2-1.9 us on this chip, a grid barrier with its agent-scope release (XCD L2 write-back) and acquire (L1 invalidate) 4-7 us,
// and the phases behind it start with cold caches.  What the head's launches cost is their own dependent chains, not the boundaries.

// The loss of a 16 x N block of logits (N <= 64), one element per thread and pass (wave 0 alone, which holds the tile, took sixteen
// divergent passes through expf / log1pf: 9 of the launch's 20 us): labels and class weights are requested with the launch's
// first batch of loads (head_loss_load), the logits come from the workgroup's LDS copy `ol` [16][N]; d out goes to the global
// matrix and to `dl` [16][N]; the block's weighted sum is added to *loss_acc by every wave.
constexpr int LOSS_U = 2;                  // 16 x 64 elements / 512 threads
struct LossRegs { float y[LOSS_U], w1[LOSS_U], w0[LOSS_U]; };
__device__ __forceinline__ LossRegs head_loss_load(const HeadLoss& L, int Bl, int N, int rb) {
    LossRegs R;
#pragma unroll
    for (int u = 0; u < LOSS_U; ++u) {
        const int e = min((int)threadIdx.x + u * HT_NT, Bl * N - 1);
        const int c = e % N;
        R.y[u] = L.labels[(size_t)rb * 16 * N + e];
        R.w1[u] = L.kind == 0 ? L.weight[2 * c] : 0.0f;
        R.w0[u] = L.kind == 0 ? L.weight[2 * c + 1] : 0.0f;
    }
    return R;
}
__device__ __forceinline__ void head_loss_block(const HeadLoss& L, const LossRegs& R, int Bl, int N, int rb, const float* ol, float inv,
                                                float scale, float* dl, double* loss_acc) {
    double part = 0.0;
#pragma unroll
    for (int u = 0; u < LOSS_U; ++u) {
        const int e = (int)threadIdx.x + u * HT_NT;
        if (e < Bl * N) {
            const float xi = ol[e], yi = R.y[u];
            float d;
            if (L.kind == 0) {                           // weighted BCE with logits, missing labels skipped (loss.hip bce_loss_kernel)
                const float wi = yi == 1.0f ? R.w1[u] : (yi == 0.0f ? R.w0[u] : 0.0f);
                part += (double)(wi * (fmaxf(xi, 0.0f) - xi * yi + log1pf(expf(-fabsf(xi)))));
                d = wi * (1.0f / (1.0f + expf(-xi)) - yi) * inv;
            } else {                                     // mean squared error (loss.hip mse_loss_kernel)
                const float df = xi - yi;
                part += (double)df * (double)df;
                d = 2.0f * df * inv;
            }
            if (L.scale) d *= scale;
            L.dout[(size_t)rb * 16 * N + e] = d;
            dl[e] = d;
        }
    }
    part = wave_sum(part);
    if ((threadIdx.x & 63) == 0 && part != 0.0) atomicAdd(loss_acc, part);
}

// Everything the row block needs is requested in ONE batch at the start -- bn2's sums, the block's 16 rows of h2, the whole of
// W3, labels, class weights, the label count -- and copied to LDS in the layouts the global matrices have; the two tiles then run
// the SAME bodies as the separate launches (hf_tile / hb_tile on descriptors that point into the workgroup's copies, rows
// renumbered from 0), d out goes from the loss to dense 3's product through LDS.  Three dependent chains (table, operand batch,
// epilogue -- each behind a launch) become one: 23 -> 9 us at configs[1].
template <bool VEC3>
__global__ __launch_bounds__(HT_NT) void head_mid_kernel(HeadMid a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int K = a.f3.K, N = a.f3.N, Kp = (K + 3) & ~3, Np = (N + 3) & ~3;
    f32x4* red = reinterpret_cast<f32x4*>(lds);                         // 32 KB: the waves' partial tiles
    float* tabA = lds + HT_RED;                                         // bn2's table, all four rows (the saved copy is written by
    float* tabB = tabA + 4 * Kp;                                        //  workgroup 0 in THIS launch); dense 3's (1, 0, 0) table
    float* xs = tabB + 3 * Np;                                          // [16][K]   rows rb*16 .. of h2 (clamped to the batch)
    float* Ws = xs + 16 * Kp;                                           // [K][N]    W3 (K*N rounded up to 4)
    float* dl = Ws + ((K * N + 3) & ~3);                                // [16][N]   d out of the block
    float* ol = dl + 16 * Np;                                           // [16][N]   logits of the block
    const unsigned* words = reinterpret_cast<const unsigned*>(a.ws);    // [1] labelled entries (HeadFwd.lab_cnt of dense 2)
    double* loss_acc = a.ws + 1;
    const int rb = blockIdx.x, B = a.f3.B;                              // one 16-row block per workgroup
    const int Bl = min(16, B - rb * 16);
    // ---- one batch of loads
    const bool vx = (K & 3) == 0 && (reinterpret_cast<uintptr_t>(a.f3.x) & 15) == 0;
    const bool vw = ((K * N) & 3) == 0 && (reinterpret_cast<uintptr_t>(a.f3.W) & 15) == 0;
    constexpr int XU = 2, WU = 2;                                       // float4 (or scalar) slots per thread and pass
    for (int p0 = 0; p0 < 16 * Kp / 4; p0 += HT_NT * XU) {                // h2 rows (K <= 2048: a few passes at most)
        f32x4 v[XU];
#pragma unroll
        for (int u = 0; u < XU; ++u) {
            const int i = min(p0 + u * HT_NT + (int)threadIdx.x, 16 * Kp / 4 - 1);
            const int r = i / (Kp / 4), c4 = i - r * (Kp / 4);
            const float* src = a.f3.x + (size_t)min(rb * 16 + r, B - 1) * K;
            if (vx) v[u] = *reinterpret_cast<const f32x4*>(src + 4 * c4);
            else {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[u][e] = src[min(4 * c4 + e, K - 1)];
            }
        }
#pragma unroll
        for (int u = 0; u < XU; ++u) {
            const int i = p0 + u * HT_NT + (int)threadIdx.x;
            if (i < 16 * Kp / 4) {
                const int r = i / (Kp / 4), c4 = i - r * (Kp / 4);
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (4 * c4 + e < K) xs[r * K + 4 * c4 + e] = v[u][e];
            }
        }
    }
    const int nw4 = (K * N + 3) >> 2;
    for (int p0 = 0; p0 < nw4; p0 += HT_NT * WU) {
        f32x4 v[WU];
#pragma unroll
        for (int u = 0; u < WU; ++u) {
            const int i = min(p0 + u * HT_NT + (int)threadIdx.x, nw4 - 1);
            if (vw) v[u] = *reinterpret_cast<const f32x4*>(a.f3.W + 4 * i);
            else {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[u][e] = a.f3.W[min(4 * i + e, K * N - 1)];
            }
        }
#pragma unroll
        for (int u = 0; u < WU; ++u) {
            const int i = p0 + u * HT_NT + (int)threadIdx.x;
            if (i < nw4) *reinterpret_cast<f32x4*>(Ws + 4 * i) = v[u];
        }
    }
    const LossRegs R = head_loss_load(a.L, Bl, N, rb);
    const float cnt = a.L.kind == 0 ? (float)words[1] : (float)(B * N);
    const float scale = a.L.scale ? *a.L.scale : 1.0f;
    hf_table(a.f3, tabA, rb == 0, true);
    hb_table(a.b3, tabB, false);
    const float inv = 1.0f / cnt;
    __syncthreads();
    // ---- the block's descriptors: rows renumbered from 0, operands in LDS
    HeadFwd f = a.f3;
    f.B = Bl; f.x = xs; f.W = Ws; f.y = a.f3.y + (size_t)rb * 16 * N; f.y2 = nullptr; f.st_out = nullptr;
    HeadBwd g = a.b3;
    g.B = Bl; g.x = xs; g.W = Ws; g.dy = dl; g.dyp = a.b3.dyp + (size_t)rb * 16 * K;
    f32x4 acc[4];
    hf_tile<VEC3, false>(f, 0, tabA, red, acc);
    if (threadIdx.x < 64) {
        hf_store(f, 0, acc);
        const int li = threadIdx.x & 15, q = threadIdx.x >> 4;         // D layout of tile j: column 4 li + j, rows 4q + r
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (4 * li + j < N) ol[(4 * q + r) * N + 4 * li + j] = acc[j][r];
    }
    __syncthreads();
    head_loss_block(a.L, R, Bl, N, rb, ol, inv, scale, dl, loss_acc);
    __syncthreads();                                                    // d out of these rows is in LDS
    const int nkb = (K + 63) >> 6;
    for (int kb = 0; kb < nkb; ++kb) {
        hb_tile<VEC3, false, 1>(g, kb, tabB, red, tabA, Kp);
        __syncthreads();
    }
}

// dense 2's backward launch with dense 3's weight gradient riding along: tiles of `a` first, then the (b) tiles of `e`
// (and the loss value published from the row blocks' partial sums: workgroup 0, HeadLossFin)
template <bool VEC, bool DROP, bool VEC3>
__global__ __launch_bounds__(HT_NT) void head_bwd_pair_kernel(HeadBwd a, HeadBwd e, HeadLossFin lf) {
    if (lf.loss && blockIdx.x == 0 && threadIdx.x == 0) {
        const double cnt = lf.kind == 0 ? (double)(float)reinterpret_cast<const unsigned*>(lf.ws)[1] : (double)(float)lf.n;
        float l = (float)(lf.ws[1] / cnt);
        if (lf.scale) l *= *lf.scale;
        lf.loss[0] = l;
    }
    extern __shared__ __attribute__((aligned(16))) float tab[];          // [3][Np] of a, [3][Np] of e, then the partial tiles
    const int Np = (a.N + 3) & ~3, Ne = (e.N + 3) & ~3;
    const int ta = hb_tiles_a(a) + hb_tiles_b(a);
    f32x4* red = reinterpret_cast<f32x4*>(tab + 3 * Np + 3 * Ne);
    if ((int)blockIdx.x < ta) {
        hb_tile<VEC, DROP>(a, blockIdx.x, tab, red, a.bnp, a.K, [&]() {
            hb_table(a, tab, blockIdx.x == 0);
            __syncthreads();
        });
    } else {
        hb_tile<VEC3, false>(e, hb_tiles_a(e) + ((int)blockIdx.x - ta), tab + 3 * Np, red, e.bnp, e.K, [&]() {
            hb_table(e, tab + 3 * Np, false);
            __syncthreads();
        });
    }
}

// column sums (sum g, sum g^2) of the read-out: 16 molecules per workgroup pre-reduced, then fp64 atomics
__global__ __launch_bounds__(256) void head_colstats_kernel(const float* __restrict__ g, int B, int F, double* __restrict__ st,
                                                             double* c0, double* c1, double* c2) {
    if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) {
        if (c0) *c0 = (double)B;
        if (c1) *c1 = (double)B;
        if (c2) *c2 = (double)B;
    }
    // grid (ceil(F/64), ceil(B/64)): lane = column, the four waves take rows r0 + wave, +4, ...
    __shared__ double part[4][64][2];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int f = blockIdx.x * 64 + lane;
    const int r0 = blockIdx.y * 64;
    double s1 = 0.0, s2 = 0.0;
    if (f < F)
        for (int r = r0 + wave; r < min(r0 + 64, B); r += 4) {
            const float v = g[(size_t)r * F + f];
            s1 += (double)v;
            s2 += (double)v * (double)v;
        }
    part[wave][lane][0] = s1;
    part[wave][lane][1] = s2;
    __syncthreads();
    if (wave == 0 && f < F) {
        s1 = (part[0][lane][0] + part[1][lane][0]) + (part[2][lane][0] + part[3][lane][0]);
        s2 = (part[0][lane][1] + part[1][lane][1]) + (part[2][lane][1] + part[3][lane][1]);
        atomicAdd(&st[2 * f], s1);
        atomicAdd(&st[2 * f + 1], s2);
    }
}

// ---- launchers --------------------------------------------------------------------------------------------------------
int head_colstats(const float* g, int B, int F, double* st, hipStream_t s, double* cnt0, double* cnt1, double* cnt2) {
    ProfScope ps(PROF_HEAD, s);
    head_colstats_kernel<<<dim3(cdiv(F, 64), cdiv(B, 64)), 256, 0, s>>>(g, B, F, st, cnt0, cnt1, cnt2);
    EAGCN_LAUNCH_CHECK();
    return EAGCN_OK;
}
static int head_cus() {
    static const int cus = [] {
        int d = 0, n = 0;
        if (hipGetDevice(&d) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, d) != hipSuccess || n < 1) n = 256;
        return n;
    }();
    return cus;
}
// waves per workgroup of a stage: HeadFwd.nw / HeadBwd.nw if set, else 8 while the launch is one round of one workgroup per CU
static int head_nw(int want, int tiles) {
    static const int cus = [] {
        int d = 0, n = 0;
        if (hipGetDevice(&d) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, d) != hipSuccess || n < 1) n = 256;
        return n;
    }();
    if (want == 4 || want == 8) return want;
    static const int env = [] { const char* e = getenv("EAGCN_HEAD_NW"); return e ? atoi(e) : 0; }();
    if (env == 4 || env == 8) return env;
    // (round 6: eight waves up to TWO workgroups per CU -- dense 1's backward at 256 molecules is 352 tiles -- measured -1 % at
    //  configs[1] while the side stream's index build was held back behind the forward, and +1.6 % (0.366 -> 0.372 ms) once it was
    //  not: one round it stays.  EAGCN_HEAD_NW = 4 | 8 | 8 x rounds + 1)
    return tiles <= cus * (env > 8 ? env / 8 : 1) ? 8 : 4;
}
int head_fwd(const HeadFwd& a, hipStream_t s) {
    EAGCN_CHECK_ARG(a.st_copies >= 1 && a.st_copies <= HEAD_COPIES, "head: %d replicas of the BatchNorm sums", a.st_copies);
    const int tiles = cdiv(a.B, 16) * cdiv(a.N, 64);
    const int nw = head_nw(a.nw, tiles);
    const size_t lds = (size_t)(2 * ((a.K + 3) & ~3) + nw * 1024) * sizeof(float);
    EAGCN_CHECK_ARG(lds <= 64 * 1024, "head: %d input features exceed the table size", a.K);
    ProfScope ps(PROF_HEAD, s, 2.0 * a.B * a.K * a.N);
    const bool vec = (a.K & 3) == 0 && (a.N & 3) == 0 && ((reinterpret_cast<uintptr_t>(a.x) | reinterpret_cast<uintptr_t>(a.W)) & 15) == 0;
#define EAGCN_HF(V, D) do { if (nw == 8) head_fwd_kernel<V, D, 8><<<tiles, 512, lds, s>>>(a); else head_fwd_kernel<V, D, 4><<<tiles, 256, lds, s>>>(a); } while (0)
    if (vec && a.drop.on) EAGCN_HF(true, true);
    else if (vec) EAGCN_HF(true, false);
    else if (a.drop.on) EAGCN_HF(false, true);
    else EAGCN_HF(false, false);
#undef EAGCN_HF
    EAGCN_LAUNCH_CHECK();
    return EAGCN_OK;
}
int head_bwd(const HeadBwd& a, hipStream_t s) {
    EAGCN_CHECK_ARG(a.ks >= 1 && a.ks <= 16 && (a.ks == 1 || a.dW_part), "head: %d row chunks of the weight gradient", a.ks);
    const int tiles = cdiv(a.B, 16) * cdiv(a.K, 64) + cdiv(a.K, 16) * cdiv(a.N, 64) * a.ks;
    const int nw = head_nw(a.nw, tiles);
    const size_t lds = (size_t)(3 * ((a.N + 3) & ~3) + nw * 1024) * sizeof(float);
    EAGCN_CHECK_ARG(lds <= 64 * 1024, "head: %d output features exceed the table size", a.N);
    ProfScope ps(PROF_HEAD, s, 4.0 * a.B * a.K * a.N);
    const uintptr_t al = reinterpret_cast<uintptr_t>(a.dy) | reinterpret_cast<uintptr_t>(a.W) | reinterpret_cast<uintptr_t>(a.y) |
                         reinterpret_cast<uintptr_t>(a.extra);
    const bool vec = (a.N & 3) == 0 && (al & 15) == 0;
    static const int light_env = [] { const char* e = getenv("EAGCN_HEAD_LIGHT"); return e ? atoi(e) : 1; }();     // rounds (of 2 workgroups per CU) from which the light kernel is taken; 0: never
    const bool light = light_env > 0 && nw == 4 && a.nw == 0 && tiles > 2 * light_env * head_cus();
#define EAGCN_HB(V, D) do { if (nw == 8) head_bwd_kernel<V, D, 8><<<tiles, 512, lds, s>>>(a); else if (light && V) head_bwd_light_kernel<D><<<tiles, 256, lds, s>>>(a); \
                            else head_bwd_kernel<V, D, 4><<<tiles, 256, lds, s>>>(a); } while (0)
    if (vec && a.drop.on) EAGCN_HB(true, true);
    else if (vec) EAGCN_HB(true, false);
    else if (a.drop.on) EAGCN_HB(false, true);
    else EAGCN_HB(false, false);
#undef EAGCN_HB
    EAGCN_LAUNCH_CHECK();
    return EAGCN_OK;
}
// the fused middle launch needs a row block of the logits to be one tile
bool head_mid_ok(int n2, int nclass) {
    static const bool env = [] { const char* v = getenv("EAGCN_HEAD_FUSED"); return !(v && v[0] == '0'); }();
    const size_t lds = (size_t)(HT_RED + 20 * ((n2 + 3) & ~3) + 35 * ((nclass + 3) & ~3) + n2 * nclass + 4) * sizeof(float);
    return env && nclass <= 64 && lds <= 150 * 1024;
}
int head_mid(const HeadMid& a, hipStream_t s) {
    const int B = a.f3.B, n2 = a.f3.K, nc = a.f3.N;
    EAGCN_CHECK_ARG(head_mid_ok(n2, nc) && a.ws && a.L.labels && a.L.loss && a.L.dout, "head_mid: %d classes / null buffer", nc);
    const uintptr_t al = reinterpret_cast<uintptr_t>(a.f3.x) | reinterpret_cast<uintptr_t>(a.f3.W) | reinterpret_cast<uintptr_t>(a.L.dout);
    const bool vec3 = (n2 & 3) == 0 && (nc & 3) == 0 && (al & 15) == 0;
    const int Kp = (n2 + 3) & ~3, Np = (nc + 3) & ~3;
    const size_t lds = (size_t)(HT_RED + 4 * Kp + 3 * Np + 16 * Kp + ((n2 * nc + 3) & ~3) + 32 * Np) * sizeof(float);
    static bool attr = [] {
        bool ok = hipFuncSetAttribute((const void*)head_mid_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 512) == hipSuccess;
        return hipFuncSetAttribute((const void*)head_mid_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 512) == hipSuccess && ok;
    }();
    (void)attr;
    EAGCN_CHECK_ARG(lds <= 150 * 1024, "head_mid: %d x %d weights exceed LDS", n2, nc);
    ProfScope ps(PROF_HEAD, s, 4.0 * B * n2 * nc);
    if (vec3) head_mid_kernel<true><<<cdiv(B, 16), HT_NT, lds, s>>>(a);
    else head_mid_kernel<false><<<cdiv(B, 16), HT_NT, lds, s>>>(a);
    EAGCN_LAUNCH_CHECK();
    return EAGCN_OK;
}
// backward of one dense layer (a) with the weight gradient of another (e: only its (b) tiles) in one grid
int head_bwd_pair(const HeadBwd& a, const HeadBwd& e, const HeadLossFin& lf, hipStream_t s) {
    EAGCN_CHECK_ARG(a.ks >= 1 && a.ks <= 16 && (a.ks == 1 || a.dW_part) && e.ks >= 1 && e.ks <= 16 && (e.ks == 1 || e.dW_part),
                    "head: %d / %d row chunks of the weight gradient", a.ks, e.ks);
    const int tiles = cdiv(a.B, 16) * cdiv(a.K, 64) + cdiv(a.K, 16) * cdiv(a.N, 64) * a.ks + cdiv(e.K, 16) * cdiv(e.N, 64) * e.ks;
    const size_t lds = (size_t)(3 * (((a.N + 3) & ~3) + ((e.N + 3) & ~3)) + HT_RED) * sizeof(float);
    EAGCN_CHECK_ARG(lds <= 64 * 1024, "head: %d output features exceed the table size", a.N);
    ProfScope ps(PROF_HEAD, s, 4.0 * a.B * a.K * a.N + 2.0 * e.B * e.K * e.N);
    const uintptr_t al = reinterpret_cast<uintptr_t>(a.dy) | reinterpret_cast<uintptr_t>(a.W) | reinterpret_cast<uintptr_t>(a.y) |
                         reinterpret_cast<uintptr_t>(a.extra);
    const bool vec = (a.N & 3) == 0 && (al & 15) == 0;
    const bool vec3 = (e.N & 3) == 0 && (reinterpret_cast<uintptr_t>(e.dy) & 15) == 0;
    const bool drop = a.drop.on != 0;
#define EAGCN_HBP(V, D, V3) head_bwd_pair_kernel<V, D, V3><<<tiles, HT_NT, lds, s>>>(a, e, lf)
    if (vec) { if (drop) { if (vec3) EAGCN_HBP(true, true, true); else EAGCN_HBP(true, true, false); }
               else { if (vec3) EAGCN_HBP(true, false, true); else EAGCN_HBP(true, false, false); } }
    else { if (drop) { if (vec3) EAGCN_HBP(false, true, true); else EAGCN_HBP(false, true, false); }
           else { if (vec3) EAGCN_HBP(false, false, true); else EAGCN_HBP(false, false, false); } }
#undef EAGCN_HBP
    EAGCN_LAUNCH_CHECK();
    return EAGCN_OK;
}
int head_gbn_bwd(const HeadGbn& a, hipStream_t s) {
    const size_t total = (size_t)a.B * a.F;
    ProfScope ps(PROF_HEAD, s);
    head_gbn_bwd_kernel<<<(int)std::max<size_t>(1, std::min<size_t>((total + 255) / 256, 1024)), 256, 0, s>>>(a);
    EAGCN_LAUNCH_CHECK();
    return EAGCN_OK;
}

}  // namespace eagcn
