// Molecule read-out: g[b,:] = sum over ALL N padded rows of x[b,i,:]  (reference models.py:108),
// optionally divided by the atom count (molfp_mode 'ave', models.py:109-111), and its backward.
// Rows that are not stored in the packed layout are all equal to `pad_row` (zero for a Concate
// layer, a constant vector for Weighted_sum, which has no mask: layers.py:315-316), so they
// contribute (N - nat[b]) * pad_row.
#include <algorithm>

#include "common.h"
#include "kernels.h"

namespace eagcn {

__device__ __forceinline__ int exact_to_packed(const ColMapD& m, int ce) {
    int eo = 0, po = 0;
    for (int s = 0; s < m.nseg; ++s) {
        if (ce < eo + m.w[s]) return po + (ce - eo);
        eo += m.w[s];
        po += m.p[s];
    }
    return -1;
}

// ---- dropout of the rows that are not stored (Weighted_sum has no row mask: reference layers.py:94 drops every element
// of the padded [B,N,F] tensor independently, layers.py:315-316 sums the views, models.py:108 sums all N rows) ------------
// A non-stored row of view k holds relu(shift_k[c]) in column c before dropout, so the read-out only needs HOW MANY of the
// m = N - nat[b] non-stored rows keep (molecule b, view k, column c): cnt[b][k][c] ~ Binomial(m, q), q the keep probability of
// the stored rows' 16-bit draws.  Round 3 drew the m Bernoulli variables (four per 64-bit hash: 50 hashes per count at the HIV
// set's N = 222 -- 0.43 ms per step, bound by the hash's 64-bit multiplies).  Now: ONE 32-bit uniform per count, inverted
// through the binomial CDF of its m -- a table of 32-bit thresholds t[m][j] = floor(2^32 P(X <= j)) built per forward by
// `pad_binomial_table_kernel` (one thread per m; plain IEEE double operations in a fixed order, so that the test suite rebuilds
// the same table in numpy: tests/test_gpu_parity.py _pad_counts) -- cnt = #{j < m : t[m][j] <= u}.  Kept for the backward pass.
// padc[b][c] = sum_k a_k relu(shift_k[c]) cnt / (1-p).
struct PadSample {
    int K, ld, fp;                           // views, output columns (one view's padded width), BatchNorm table stride
    int off[EAGCN_MAX_VIEWS];                // column offset of view k inside the [fp] tables
    const float* bn_sh;                      // [fp] BatchNorm shift (beta - mean * scale)
    const float* ave_w;                      // [K]
    uint64_t seed; const uint64_t* seed_dev;
    uint32_t thr16; float inv_keep;
    uint16_t* cnt;                           // [B][K][ld]
    float* padc;                             // [B][ld]
    const uint32_t* tab;                     // [N + 1][N + 1] thresholds (row m: entries 0 .. m)
    int tab_ld;
};
// One workgroup per m, one thread per outcome k.  Weights relative to the mode (no underflow for any m): w[k] for k >= mode is the
// product of the ratios w[j+1] / w[j] = (m - j) r / (j + 1) from the mode up, for k < mode the product of their inverses from the
// mode down; then the running sum.  The three scans are Hillis-Steele scans (step d = 1, 2, 4, ...: x[i] <- x[i] op x[i -+ d], all from
// the previous step's values), every operation a correctly rounded IEEE double operation issued explicitly (no fused multiply-add):
// a fixed order that tests/helpers.py pad_thresholds repeats in numpy.  (A first version walked each row sequentially in one thread:
// 151 us at N = 222 -- hundreds of dependent double divisions.)
__global__ __launch_bounds__(1024) void pad_binomial_table_kernel(eagcn_batch bt, uint32_t thr16, uint32_t* __restrict__ tab, int tab_ld) {
    __shared__ double x_s[2][1024];
    const int m = blockIdx.x, k = threadIdx.x;
    if (m > dev_n(bt) || m >= tab_ld) return;           // (uniform for the workgroup)
    const double q = __dsub_rn(1.0, __ddiv_rn((double)thr16, 65536.0));           // keep probability of a 16-bit draw
    const double r = __ddiv_rn(q, __dsub_rn(1.0, q));
    int mode = (int)floor(__dmul_rn((double)(m + 1), q));
    mode = mode < 0 ? 0 : (mode > m ? m : mode);
    const bool in = k <= m;
    // scan helper: `down` = suffix scan (x[i] op x[i + d]), else prefix scan (x[i] op x[i - d]); mul or add
    auto scan = [&](double v, bool down, bool mul) -> double {
        int cur = 0;
        x_s[0][k] = v;
        __syncthreads();
        for (int d = 1; d <= m; d <<= 1) {
            const int o = down ? k + d : k - d;
            double y = x_s[cur][k];
            if (in && o >= 0 && o <= m) y = mul ? __dmul_rn(y, x_s[cur][o]) : __dadd_rn(y, x_s[cur][o]);
            x_s[cur ^ 1][k] = y;
            __syncthreads();
            cur ^= 1;
        }
        const double res = x_s[cur][k];
        __syncthreads();
        return res;
    };
    // up[k] = ratio w[k] / w[k-1] for k > mode, else 1;  dn[k] = w[k] / w[k+1] for k < mode, else 1
    double up = 1.0, dn = 1.0;
    if (in && k > mode) up = __ddiv_rn(__dmul_rn((double)(m - k + 1), r), (double)k);
    if (in && k < mode) dn = __ddiv_rn((double)(k + 1), __dmul_rn((double)(m - k), r));
    const double pu = scan(up, false, true), pd = scan(dn, true, true);
    const double w = k >= mode ? pu : pd;
    const double c = scan(in ? w : 0.0, false, false);   // running sum; the total is its last entry
    __shared__ double S_s;
    if (k == m) S_s = c;
    __syncthreads();
    if (in) {
        const double y = __dmul_rn(__ddiv_rn(c, S_s), 4294967296.0);
        tab[(size_t)m * tab_ld + k] = y >= 4294967295.0 ? 0xFFFFFFFFu : (uint32_t)y;
    }
}
__global__ __launch_bounds__(256) void readout_pad_sample_kernel(eagcn_batch bt, PadSample a) {
    __shared__ uint32_t t_s[1025];
    const int b = blockIdx.y;
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t seed = a.seed_dev ? *a.seed_dev : a.seed;
    const int m = max(dev_n(bt) - bt.nat[b], 0);         // non-stored rows of this molecule
    for (int j = threadIdx.x; j < m; j += blockDim.x) t_s[j] = a.tab[(size_t)m * a.tab_ld + j];
    __syncthreads();
    if (c >= a.ld) return;
    float acc = 0.0f;
    for (int k = 0; k < a.K; ++k) {
        const int cp = a.off[k] + c;
        const uint32_t u = (uint32_t)(rng_u64(seed, (PAD_STREAM_BASE + (uint64_t)b) * (uint64_t)a.fp + (uint64_t)cp) >> 32);
        int lo = 0, hi = m;                              // cnt = first j with t[j] > u (t is non-decreasing)
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (t_s[mid] <= u) lo = mid + 1; else hi = mid;
        }
        a.cnt[((size_t)b * a.K + k) * a.ld + c] = (uint16_t)lo;
        acc += a.ave_w[k] * fmaxf(a.bn_sh[cp], 0.0f) * ((float)lo * a.inv_keep);
    }
    a.padc[(size_t)b * a.ld + c] = acc;
}
// d(value of the non-stored rows of view k, column c) = sum_b cnt[b][k][c] / (1-p) * dg[b][c] / size[b]
// grid (ceil(ld / 64), K), 1024 threads: lane = packed column, the sixteen waves split the molecules -- eight molecules'
// loads in flight per wave, every wave's share and the sixteen partial sums added in a fixed order (a single thread per column
// walking all B molecules was a 1024-deep chain of dependent loads on 25 workgroups: 299 us at B = 1024, ld = 1250)
__global__ __launch_bounds__(1024) void readout_bwd_pad_views_kernel(eagcn_batch bt, const float* __restrict__ dg, ColMapD m,
                                                                      int ld, const int64_t* __restrict__ size, int mode, int F,
                                                                      int K, const uint16_t* __restrict__ cnt, float inv_keep,
                                                                      float* __restrict__ dpad) {
    __shared__ double part[16][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int cp = blockIdx.x * 64 + lane, k = blockIdx.y;
    int eo = 0, po = 0, ce = -1;
    if (cp < ld)
        for (int s = 0; s < m.nseg; ++s) {
            if (cp < po + m.p[s]) { ce = (cp - po < m.w[s]) ? eo + (cp - po) : -1; break; }
            eo += m.w[s];
            po += m.p[s];
        }
    double acc = 0.0;
    if (ce >= 0) {
        const int per = (bt.B + 15) >> 4;
        const int b0 = wave * per, b1 = min(bt.B, b0 + per);
        for (int b = b0; b < b1; b += 8) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int bb = min(b + u, b1 - 1);
                const float inv = mode == 1 ? 1.0f / (float)size[bb] : 1.0f;
                const float t = (float)cnt[((size_t)bb * K + k) * ld + cp] * inv_keep * dg[(size_t)bb * F + ce] * inv;
                v[u] = b + u < b1 ? t : 0.0f;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) acc += (double)v[u];
        }
    }
    part[wave][lane] = acc;
    __syncthreads();
    if (wave == 0 && cp < ld) {
        double t = 0.0;
#pragma unroll
        for (int w = 0; w < 16; ++w) t += part[w][lane];
        dpad[(size_t)k * ld + cp] = (float)t;
    }
}

// grid (B, ceil(F/64)); 4 wavefronts split the molecule's rows, 64 lanes own 64 exact columns.
// padc (optional): per-molecule contribution of the non-stored rows (sampled dropout), replaces (N - nat) * pad_row
__global__ __launch_bounds__(256) void readout_fwd_kernel(eagcn_batch bt, const float* __restrict__ x, ColMapD m,
                                                           int ld, const float* __restrict__ pad_row,
                                                           const float* __restrict__ padc,
                                                           const int64_t* __restrict__ size, int mode,
                                                           float* __restrict__ g, int F) {
    __shared__ float part[4][64];
    const int b = blockIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int f = blockIdx.y * 64 + lane;
    const int n = bt.nat[b], r0 = bt.row0[b];
    const int cp = f < F ? exact_to_packed(m, f) : 0;
    float s = 0.0f;
    if (f < F)
        for (int i0 = wave; i0 < n; i0 += 16) {           // rows i0, i0+4, i0+8, i0+12 loaded together
            float v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int i = i0 + 4 * u;
                v[u] = i < n ? x[(size_t)(r0 + i) * ld + cp] : 0.0f;
            }
            s += (v[0] + v[1]) + (v[2] + v[3]);
        }
    part[wave][lane] = s;
    __syncthreads();
    if (wave == 0 && f < F) {
        s = (part[0][lane] + part[1][lane]) + (part[2][lane] + part[3][lane]);
        if (padc) s += padc[(size_t)b * ld + cp];
        else if (pad_row) s += (float)(dev_n(bt) - n) * pad_row[cp];
        const float inv = mode == 1 ? 1.0f / (float)size[b] : 1.0f;
        g[(size_t)b * F + f] = s * inv;
    }
}

// Read-out straight from the pre-BatchNorm matrix of a Concate top layer: x = mask * dropout(relu(bn(y))) (layers.py:93-94, 313) is
// formed on the operand load and summed over the atoms, and the column sums of g that Graph_BN needs (models.py:112) are taken
// by the same launch -- three launches (bn_apply of the top layer, readout_fwd, head_colstats) and the write + re-read of the
// top layer's [T, F] output become one.  The output matrix itself is only built when somebody asks for the atom
// representations (eagcn_model_atom_rep_materialize).
// grid (ceil(B/4), ceil(F/64)): one WAVE per molecule, 64 lanes = 64 exact columns, four rows in flight.
__global__ __launch_bounds__(256) void readout_bn_fwd_kernel(eagcn_batch bt, ColMapD m, ReadoutBn a) {
    // One workgroup per FOUR consecutive molecules: their packed rows are contiguous, and the four waves take them round-robin
    // (eight rows in flight per wave) whatever molecule they belong to -- a 132-atom molecule next to three 16-atom ones costs
    // every wave a quarter of the rows instead of one wave all 132.  Each wave keeps one partial sum per molecule.
    __shared__ float part[4][4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int b0 = blockIdx.x * 4;
    const int f = blockIdx.y * 64 + lane;
    if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) {
        if (a.cnt0) *a.cnt0 = (double)bt.B;
        if (a.cnt1) *a.cnt1 = (double)bt.B;
        if (a.cnt2) *a.cnt2 = (double)bt.B;
    }
    const int nb = min(4, bt.B - b0);
    const bool okf = f < a.F;
    const int cp = okf ? exact_to_packed(m, f) : 0;
    float s[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    const int rbeg = bt.row0[b0], rend = bt.row0[b0 + nb];
    int bnd[4];                                            // first row of the next molecule, per molecule of the group
#pragma unroll
    for (int j = 0; j < 4; ++j) bnd[j] = bt.row0[min(b0 + j + 1, b0 + nb)];
    if (okf) {
        const uint64_t seed = a.do_drop ? (a.seed_dev ? *a.seed_dev : a.seed) : 0ull;
        const float sc = a.bn[BN_SC * a.fp + cp], sh = a.bn[BN_SH * a.fp + cp];
        for (int r0 = rbeg + wave; r0 < rend; r0 += 32) {
            float y[8], mk[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int r = min(r0 + 4 * u, rend - 1);
                y[u] = a.Y[(size_t)r * a.ldy + cp];
                mk[u] = bt.row_m[r];
            }
            float ds[8];
            if (a.do_drop && a.pair) {
                // one 64-bit draw serves the four adjacent columns of a lane quad (drop_scale_el): lane j of the quad draws for
                // row 4 h + j of this round's eight and every lane picks its own 16-bit field out of the four draws -- a quarter
                // of the hashes (the kernel is bound by their 64-bit multiplies)
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int rs = r0 + 4 * (4 * h + (lane & 3));
                    const uint64_t zo = rng_u64(seed, ((uint64_t)rs * a.fp + cp) >> 2);
                    const int lo = (int)(uint32_t)zo, hi = (int)(uint32_t)(zo >> 32);
                    uint32_t w[4];                         // the half of row j's draw that holds this lane's field
                    {
                        const uint32_t l0 = (uint32_t)__builtin_amdgcn_mov_dpp(lo, 0x00, 0xF, 0xF, true);
                        const uint32_t h0 = (uint32_t)__builtin_amdgcn_mov_dpp(hi, 0x00, 0xF, 0xF, true);
                        const uint32_t l1 = (uint32_t)__builtin_amdgcn_mov_dpp(lo, 0x55, 0xF, 0xF, true);
                        const uint32_t h1 = (uint32_t)__builtin_amdgcn_mov_dpp(hi, 0x55, 0xF, 0xF, true);
                        const uint32_t l2 = (uint32_t)__builtin_amdgcn_mov_dpp(lo, 0xAA, 0xF, 0xF, true);
                        const uint32_t h2 = (uint32_t)__builtin_amdgcn_mov_dpp(hi, 0xAA, 0xF, 0xF, true);
                        const uint32_t l3 = (uint32_t)__builtin_amdgcn_mov_dpp(lo, 0xFF, 0xF, 0xF, true);
                        const uint32_t h3 = (uint32_t)__builtin_amdgcn_mov_dpp(hi, 0xFF, 0xF, 0xF, true);
                        const bool up = (lane & 2) != 0;
                        w[0] = up ? h0 : l0; w[1] = up ? h1 : l1; w[2] = up ? h2 : l2; w[3] = up ? h3 : l3;
                    }
                    const int sh = 16 * (lane & 1);
#pragma unroll
                    for (int j = 0; j < 4; ++j) ds[4 * h + j] = ((w[j] >> sh) & 0xFFFFu) >= (a.thr >> 16) ? a.inv_keep : 0.0f;
                }
            } else {
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    ds[u] = a.do_drop ? drop_scale_el(seed, (uint64_t)(r0 + 4 * u) * a.fp + cp, a.thr, a.inv_keep) : 1.0f;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int r = r0 + 4 * u;
                if (r < rend) {
                    float v = fmaxf(y[u] * sc + sh, 0.0f) * mk[u];
                    if (a.do_drop) v *= ds[u];
                    const int mi = (r >= bnd[0] ? 1 : 0) + (r >= bnd[1] ? 1 : 0) + (r >= bnd[2] ? 1 : 0);
#pragma unroll
                    for (int j = 0; j < 4; ++j) s[j] += mi == j ? v : 0.0f;
                }
            }
        }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) part[wave][j][lane] = s[j];
    __syncthreads();
    if (wave == 0 && okf) {
        // (non-stored rows of a Concate layer are masked to zero: nothing to add for them)
        double s1 = 0.0, s2 = 0.0;
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (j < nb) {
                float v = (part[0][j][lane] + part[1][j][lane]) + (part[2][j][lane] + part[3][j][lane]);
                if (a.mode == 1) v *= 1.0f / (float)a.size[b0 + j];
                a.g[(size_t)(b0 + j) * a.F + f] = v;
                s1 += (double)v;
                s2 += (double)v * (double)v;
            }
        double* st = a.st + (size_t)(blockIdx.x % a.st_copies) * a.st_stride;
        atomicAdd(&st[2 * f], s1);
        atomicAdd(&st[2 * f + 1], s2);
    }
}

int readout_bn_forward(const eagcn_batch* b, const eagcn_layout* lay, const ReadoutBn& a, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    ProfScope ps(PROF_READOUT, s);
    ReadoutBn aa = a;
    // lanes 4i .. 4i+3 hold the four fields of one dropout draw when every column range starts and ends on a multiple of four
    aa.pair = (a.fp & 3) == 0 && (a.F & 3) == 0;
    for (int sg = 0; sg < lay->nseg; ++sg) aa.pair = aa.pair && (lay->width[sg] & 3) == 0 && (lay->pad[sg] & 3) == 0;
    readout_bn_fwd_kernel<<<dim3(cdiv(b->B, 4), cdiv(a.F, 64)), 256, 0, s>>>(*b, make_colmap(lay), aa);
    EAGCN_LAUNCH_CHECK();
    return EAGCN_OK;
}

// dx[r][cp] = dg[mol(r)][exact(cp)] / size: one thread per packed element, fully parallel
__global__ __launch_bounds__(256) void readout_bwd_kernel(eagcn_batch bt, const float* __restrict__ dg, ColMapD m,
                                                           int ld, const int64_t* __restrict__ size, int mode,
                                                           int F, float* __restrict__ dx) {
    const size_t total = (size_t)dev_rows(bt) * ld;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const int r = (int)(e / ld), cp = (int)(e % ld);
        int eo = 0, po = 0, ce = -1;
        for (int s = 0; s < m.nseg; ++s) {
            if (cp < po + m.p[s]) { ce = (cp - po < m.w[s]) ? eo + (cp - po) : -1; break; }
            eo += m.w[s];
            po += m.p[s];
        }
        const int b = bt.row_mol[r];
        const float inv = mode == 1 ? 1.0f / (float)size[b] : 1.0f;
        dx[e] = ce >= 0 ? dg[(size_t)b * F + ce] * inv : 0.0f;
    }
}

// d pad_row[c] = sum_b (N - nat[b]) * dg[b][c] / size[b]; grid ceil(ld / 64), 1024 threads, the molecules split over the
// sixteen waves like readout_bwd_pad_views_kernel
__global__ __launch_bounds__(1024) void readout_bwd_pad_kernel(eagcn_batch bt, const float* __restrict__ dg, ColMapD m,
                                                                int ld, const int64_t* __restrict__ size, int mode,
                                                                int F, float* __restrict__ dpad) {
    __shared__ double part[16][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int cp = blockIdx.x * 64 + lane;
    int eo = 0, po = 0, ce = -1;
    if (cp < ld)
        for (int s = 0; s < m.nseg; ++s) {
            if (cp < po + m.p[s]) { ce = (cp - po < m.w[s]) ? eo + (cp - po) : -1; break; }
            eo += m.w[s];
            po += m.p[s];
        }
    double acc = 0.0;
    if (ce >= 0) {
        const int per = (bt.B + 15) >> 4, n = dev_n(bt);
        const int b0 = wave * per, b1 = min(bt.B, b0 + per);
        for (int b = b0; b < b1; b += 8) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int bb = min(b + u, b1 - 1);
                const float inv = mode == 1 ? 1.0f / (float)size[bb] : 1.0f;
                const float t = (float)(n - bt.nat[bb]) * dg[(size_t)bb * F + ce] * inv;
                v[u] = b + u < b1 ? t : 0.0f;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) acc += (double)v[u];
        }
    }
    part[wave][lane] = acc;
    __syncthreads();
    if (wave == 0 && cp < ld) {
        double t = 0.0;
#pragma unroll
        for (int w = 0; w < 16; ++w) t += part[w][lane];
        dpad[cp] = (float)t;
    }
}

// forward read-out with the non-stored rows' dropout SAMPLED (Weighted_sum, training, p > 0): fills cnt / padc, then sums
int readout_forward_sampled(const eagcn_batch* b, const float* x, const eagcn_layout* lay, const eagcn_layer_params* p,
                            const float* bn_sh, const int64_t* size, int mode, float* g, int F, uint16_t* cnt, float* padc,
                            uint32_t* tab, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    const int ld = layout_ld(lay);
    EAGCN_CHECK_ARG(b->N <= 1023, "read-out of a Weighted_sum layer under dropout: N=%d exceeds the 1023 atom slots its kept-row table is built for", b->N);
    PadSample a;
    a.K = p->K; a.ld = ld; a.fp = p->K * ld;
    for (int k = 0; k < EAGCN_MAX_VIEWS; ++k) a.off[k] = k * ld;
    a.bn_sh = bn_sh; a.ave_w = p->ave_w; a.seed = p->seed; a.seed_dev = p->seed_dev;
    a.thr16 = (uint32_t)std::min(65535.0, (double)p->dropout * 65536.0);
    a.inv_keep = 1.0f / (1.0f - p->dropout);
    a.cnt = cnt; a.padc = padc;
    a.tab = tab; a.tab_ld = b->N + 1;
    ProfScope ps(PROF_READOUT, s);
    pad_binomial_table_kernel<<<b->N + 1, 1024, 0, s>>>(*b, a.thr16, tab, a.tab_ld);
    EAGCN_LAUNCH_CHECK();
    readout_pad_sample_kernel<<<dim3(cdiv(ld, 256), b->B), 256, 0, s>>>(*b, a);
    EAGCN_LAUNCH_CHECK();
    readout_fwd_kernel<<<dim3(b->B, cdiv(F, 64)), 256, 0, s>>>(*b, x, make_colmap(lay), ld, nullptr, padc, size, mode, g, F);
    EAGCN_LAUNCH_CHECK();
    return EAGCN_OK;
}
// its backward half: dpad [K][ld], one gradient row per view
int readout_backward_pad_views(const eagcn_batch* b, const float* dg, const eagcn_layout* lay, const int64_t* size, int mode,
                               int F, int K, const uint16_t* cnt, float dropout, float* dpad, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    ProfScope ps(PROF_READOUT, s);
    readout_bwd_pad_views_kernel<<<dim3(cdiv(layout_ld(lay), 64), K), 1024, 0, s>>>(*b, dg, make_colmap(lay), layout_ld(lay), size,
                                                                                mode, F, K, cnt, 1.0f / (1.0f - dropout), dpad);
    EAGCN_LAUNCH_CHECK();
    return EAGCN_OK;
}

int readout_backward_pad(const eagcn_batch* b, const float* dg, const eagcn_layout* lay, const int64_t* size,
                         int mode, int F, float* dpad_row, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    ProfScope ps(PROF_READOUT, s);
    readout_bwd_pad_kernel<<<cdiv(layout_ld(lay), 64), 1024, 0, s>>>(*b, dg, make_colmap(lay), layout_ld(lay), size,
                                                                   mode, F, dpad_row);
    EAGCN_LAUNCH_CHECK();
    return EAGCN_OK;
}

}  // namespace eagcn

using namespace eagcn;

extern "C" int eagcn_readout_forward(const eagcn_batch* b, const float* x, const eagcn_layout* lay,
                                     const float* pad_row, const int64_t* size, int mode, float* g, int F,
                                     void* stream) {
    hipStream_t s = (hipStream_t)stream;
    EAGCN_CHECK_ARG(b && lay && g, "eagcn_readout_forward: null argument");
    EAGCN_CHECK_ARG(b->T == 0 || x, "eagcn_readout_forward: null activations");
    EAGCN_CHECK_ARG(layout_width(lay) == F, "eagcn_readout_forward: layout width %d != F %d", layout_width(lay), F);
    EAGCN_CHECK_ARG(mode == 0 || (mode == 1 && size), "eagcn_readout_forward: mode 1 ('ave') needs size");
    ProfScope ps(PROF_READOUT, s);
    readout_fwd_kernel<<<dim3(b->B, cdiv(F, 64)), 256, 0, s>>>(*b, x, make_colmap(lay), layout_ld(lay), pad_row, nullptr, size, mode, g, F);
    EAGCN_LAUNCH_CHECK();
    return EAGCN_OK;
}

// the rows of a per-molecule read-out gradient, [T][ld] (layer.hip: upstream gradient of a Weighted_sum top layer for lagg.hip's staging
// and for the BatchNorm backward's reduction).  A thread owns ONE packed column (its exact column is looked up once) and walks rows:
// the molecule of a row is uniform for the workgroup, loads and stores are coalesced, no division (readout_bwd_kernel's flat element
// index cost a 64-bit division per element: 88 us for the 131 MB of the HIV top layer)
__global__ __launch_bounds__(256) void readout_bwd_rows_kernel(eagcn_batch bt, const float* __restrict__ dg, ColMapD m, int ld,
                                                                const int64_t* __restrict__ size, int mode, int F, float* __restrict__ dx) {
    const int cp = blockIdx.x * 256 + threadIdx.x;
    if (cp >= ld) return;
    int eo = 0, po = 0, ce = -1;
    for (int s = 0; s < m.nseg; ++s) {
        if (cp < po + m.p[s]) { ce = (cp - po < m.w[s]) ? eo + (cp - po) : -1; break; }
        eo += m.w[s];
        po += m.p[s];
    }
    const int T = dev_rows(bt);
    for (int r0 = blockIdx.y * 8; r0 < T; r0 += gridDim.y * 8) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int r = min(r0 + u, T - 1);
            const int b = bt.row_mol[r];
            const float inv = mode == 1 ? 1.0f / (float)size[b] : 1.0f;
            v[u] = ce >= 0 ? dg[(size_t)b * F + ce] * inv : 0.0f;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (r0 + u < T) dx[(size_t)(r0 + u) * ld + cp] = v[u];
    }
}
int eagcn::launch_readout_bwd_rows(const eagcn_batch* b, const ReadoutGrad& rg, int ld, float* dx, hipStream_t s) {
    const int gx = cdiv(ld, 256);
    const int gy = std::max(1, std::min(cdiv(std::max(b->T, 1), 8), std::max(1, 4096 / gx)));
    readout_bwd_rows_kernel<<<dim3(gx, gy), 256, 0, s>>>(*b, rg.dg, rg.map, ld, rg.size, rg.mode, rg.F, dx);
    EAGCN_LAUNCH_CHECK();
    return EAGCN_OK;
}

extern "C" int eagcn_readout_backward(const eagcn_batch* b, const float* dg, const eagcn_layout* lay,
                                      const int64_t* size, int mode, int F, float* dx, float* dpad_row,
                                      void* stream) {
    hipStream_t s = (hipStream_t)stream;
    EAGCN_CHECK_ARG(b && lay && dg, "eagcn_readout_backward: null argument");
    EAGCN_CHECK_ARG(b->T == 0 || dx, "eagcn_readout_backward: null dx");
    EAGCN_CHECK_ARG(layout_width(lay) == F, "eagcn_readout_backward: layout width %d != F %d", layout_width(lay), F);
    EAGCN_CHECK_ARG(mode == 0 || (mode == 1 && size), "eagcn_readout_backward: mode 1 ('ave') needs size");
    ProfScope ps(PROF_READOUT, s);
    {
        const size_t total = (size_t)std::max(b->T, 1) * layout_ld(lay);
        const int grid = (int)std::max<size_t>(1, std::min<size_t>((total + 255) / 256, 2048));
        readout_bwd_kernel<<<grid, 256, 0, s>>>(*b, dg, make_colmap(lay), layout_ld(lay), size, mode, F, dx);
    }
    EAGCN_LAUNCH_CHECK();
    if (dpad_row) {
        readout_bwd_pad_kernel<<<cdiv(layout_ld(lay), 64), 1024, 0, s>>>(*b, dg, make_colmap(lay), layout_ld(lay), size,
                                                                       mode, F, dpad_row);
        EAGCN_LAUNCH_CHECK();
    }
    return EAGCN_OK;
}
