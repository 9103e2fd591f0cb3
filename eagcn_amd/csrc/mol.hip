// Molecule-staged backward of the edge-attention aggregation: ONE kernel per layer for
//   (1) the BatchNorm-backward affine         dY' = sc (dH - c1 - xhat c2)                      (layers.py:408-412 under autograd)
//   (2) the transposed aggregation            dP[j,:] = sum_i A^[i,j] dY'[i,:]                  (layers.py:39, 87-90)
//   (3) the edge gradients                    d att.weight, d self_r  (closed form: SURVEY.md 8a, agg.hip)
//   (4) their final reduction into the parameter gradients (last workgroup, ticket)
// which used to be four launches (bn_bwd_apply, agg_edge, unpack_grads + the in-place dY' pass) that together read the
// [T, Fp] matrices dH / Y' / P about ten times (the 16-row tiles of a molecule re-read its whole slice, the edge kernel read dY'
// and Y' again and gathered P rows per bond) -- here every element of dH, Y' and P is read from memory exactly ONCE.
//
// Work item = (molecule, view, column piece): a molecule of n atoms is cut into ceil(n/16) pieces along the COLUMNS of the
// view (the index's tile table [molecule, piece, n, row0] serves as the work list, so big molecules get proportionally more
// workgroups and a piece's work grows linearly with n, not quadratically).  The workgroup stages dY' (computed on the way in)
// and P of ALL n rows x its W columns in LDS with 16-byte coalesced loads -- all of them in flight at once --, then
//   * its four wavefronts run the transposed aggregation on the matrix cores (v_mfma_f32_16x16x4_f32, A operand built on
//     the fly from the bond-type codes and the sigma table exactly as agg.hip does, 1e-9 filler included, B operand from LDS),
//   * and walk the bonds of every row for the edge gradients with both operands of <dY'_i, P_j> in LDS.  The row term
//     <dY'_i, Y'_i> is taken while staging; both are partial sums over the piece's columns, which is all the closed form needs:
//     dU_ij = rs_i (g_ij - rd_i) is linear in them.
// Bond-type histograms go to eight fp64 accumulator copies in global memory (a dozen non-zero bins per workgroup); the last
// workgroup to finish (ticket) drains them with atomic exchanges -- which also leaves them zero for the next launch -- and
// writes d att.weight / d self_r, so no reduction launch follows.
#include <algorithm>

#include <stdlib.h>

#include "common.h"
#include "kernels.h"

namespace eagcn {

constexpr int MB_HMAX = 64;                 // queued bonds per row group (flushed when full)
constexpr int MB_MAXCW = 12;                // column tiles of a piece (and of a y-chunk of the grid)

// column tiles [t0, t0 + cw) of piece j (of nc) inside y-chunk ch of a view with ntk column tiles
__host__ __device__ inline void mol_piece(int ntk, int CT, int ch, int nc, int j, int& t0, int& cw) {
    const int lo = ch * CT, hi = ntk < lo + CT ? ntk : lo + CT;
    const int ctl = hi > lo ? hi - lo : 0;
    const int a = (j * ctl) / nc, b = ((j + 1) * ctl) / nc;
    t0 = lo + a;
    cw = b - a;
}

// physical LDS index of element (row, col) of an [n][W] slice: rows are W floats apart; W = 16 (mod 32) puts consecutive rows
// on complementary bank halves by itself, W = 0 (mod 32) needs the 16-column swap on odd rows (MFMA B-operand reads take
// rows i, i+1 in the two halves of a 32-lane group)
__device__ __forceinline__ int mol_at(int row, int col, int W, bool swz) { return row * W + (swz ? (col ^ ((row & 1) << 4)) : col); }

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 8))) void mol_bwd_kernel(MolBwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    __shared__ float sig_s[256];
    __shared__ double h_s[256];
    __shared__ double dr_s[16];
    __shared__ int hit_s[16][MB_HMAX];
    __shared__ int last_s;
    const eagcn_batch& bt = a.bt;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, q = lane >> 4;
    const int grp = tid >> 4, sl = tid & 15;
    const int k = blockIdx.y / a.nchunk, ch = blockIdx.y - k * a.nchunk;
    // independent loads first: the work item (inside the tile CAPACITY: always a legal address), the device-side tile count,
    // the sigma table
    const int4 ti = reinterpret_cast<const int4*>(bt.tile_info)[blockIdx.x];
    const float sig_v = a.sig[k * 256 + tid];
    const float r = a.rsig[k];
    const int ntiles = dev_tiles(bt);
    const int fp = a.vc.off[a.vc.K];
    bool work = (int)blockIdx.x < ntiles;
    const int b = ti.x, pj = ti.y, n = work ? ti.z : 0, r0 = ti.w;
    int t0 = 0, cw = 0;
    if (work) {
        mol_piece((a.vc.off[k + 1] - a.vc.off[k]) / 16, a.CT, ch, (n + 15) >> 4, pj, t0, cw);
        work = cw > 0;
    }
    if (work) {
        const int W = cw * 16, nf4 = W >> 2;
        const bool swz = (W & 31) == 0;
        const int c0 = a.vc.off[k] + t0 * 16;
        float* dYs = smem;
        float* Ps = smem + (size_t)n * W;
        float* rd_s = Ps + (size_t)n * W;
        float* rs_s = rd_s + n;
        const uint8_t* codeb = bt.code + ((size_t)k * bt.B + b) * bt.N * bt.ldc;
        sig_s[tid] = sig_v;
        h_s[tid] = 0.0;
        if (tid < 16) dr_s[tid] = 0.0;
        for (int i = tid; i < n; i += 256) rs_s[i] = a.rscale[(size_t)k * bt.T + r0 + i];

        // ---- phase 1 prefetch: bond-type codes of this wave's first aggregation unit (64 source rows) ---------------------
        const int RT = (n + 15) >> 4;
        const int CG = RT >= 4 ? 1 : (RT == 1 ? min(cw, 4) : min(cw, 2));
        const int nunits = RT * CG;
        // (16 single-byte loads -- the codes of column ja in 16 source rows -- packed four to a register)
        auto load_codes = [&](int rt, int ibase, uint32_t (&cc)[4]) __attribute__((always_inline)) {
            const int ja = rt * 16 + li;
            uint32_t v[16];
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int i = ibase + 4 * e + q;                     // k-step e of the block: source row i
                v[e] = (i < n && ja < n) ? (uint32_t)codeb[(size_t)i * bt.ldc + ja] : 0u;
            }
#pragma unroll
            for (int w4 = 0; w4 < 4; ++w4) cc[w4] = v[4 * w4] | (v[4 * w4 + 1] << 8) | (v[4 * w4 + 2] << 16) | (v[4 * w4 + 3] << 24);
        };
        uint32_t cpre[4] = {0u, 0u, 0u, 0u};
        if (wave < nunits) load_codes(wave / CG, 0, cpre);
        // ---- phase 2 prefetch: the code row of this group's first row ------------------------------------------------------
        uint4 crow_pre = make_uint4(0u, 0u, 0u, 0u);
        if (grp < n && sl * 16 < n) crow_pre = *reinterpret_cast<const uint4*>(codeb + (size_t)grp * bt.ldc + sl * 16);

        // ---- phase 0: stage dY' and P of all n rows x W columns; row term <dY'_i, Y'_i> over these columns -----------------
        {
            // dY' = sc (dH - c1 - (y - mu) inv c2) = al dH + be y + ga per column (three coefficient vectors in registers
            // instead of five: the kernel is held at 128 VGPRs)
            float4 cal[3], cbe[3], cga[3];
#pragma unroll
            for (int u = 0; u < 3; ++u) {
                const int col = c0 + 4 * min(sl + 16 * u, nf4 - 1);
                const float4 sc4 = *reinterpret_cast<const float4*>(a.bn + BN_SC * fp + col);
                const float4 mu4 = *reinterpret_cast<const float4*>(a.bn + BN_MU * fp + col);
                const float4 iv4 = *reinterpret_cast<const float4*>(a.bn + BN_INV * fp + col);
                const float4 c14 = *reinterpret_cast<const float4*>(a.cc + col);
                const float4 c24 = *reinterpret_cast<const float4*>(a.cc + fp + col);
                cal[u] = sc4;
                cbe[u] = make_float4(-sc4.x * iv4.x * c24.x, -sc4.y * iv4.y * c24.y, -sc4.z * iv4.z * c24.z, -sc4.w * iv4.w * c24.w);
                cga[u] = make_float4(sc4.x * (iv4.x * c24.x * mu4.x - c14.x), sc4.y * (iv4.y * c24.y * mu4.y - c14.y),
                                     sc4.z * (iv4.z * c24.z * mu4.z - c14.z), sc4.w * (iv4.w * c24.w * mu4.w - c14.w));
            }
            for (int i = grp; i < n; i += 16) {
                const size_t ro = (size_t)(r0 + i) * a.ld + c0;
                float4 dh[3], yy[3], pp[3];
#pragma unroll
                for (int u = 0; u < 3; ++u) {
                    const int cidx = min(sl + 16 * u, nf4 - 1);
                    dh[u] = *reinterpret_cast<const float4*>(a.dH + ro + 4 * cidx);
                    yy[u] = *reinterpret_cast<const float4*>(a.Y + ro + 4 * cidx);
                    pp[u] = *reinterpret_cast<const float4*>(a.P + ro + 4 * cidx);
                }
                float part = 0.0f;
#pragma unroll
                for (int u = 0; u < 3; ++u) {
                    const int cidx = sl + 16 * u;
                    if (cidx < nf4) {
                        float4 d;
                        d.x = fmaf(cal[u].x, dh[u].x, fmaf(cbe[u].x, yy[u].x, cga[u].x));
                        d.y = fmaf(cal[u].y, dh[u].y, fmaf(cbe[u].y, yy[u].y, cga[u].y));
                        d.z = fmaf(cal[u].z, dh[u].z, fmaf(cbe[u].z, yy[u].z, cga[u].z));
                        d.w = fmaf(cal[u].w, dh[u].w, fmaf(cbe[u].w, yy[u].w, cga[u].w));
                        *reinterpret_cast<float4*>(dYs + mol_at(i, 4 * cidx, W, swz)) = d;
                        *reinterpret_cast<float4*>(Ps + mol_at(i, 4 * cidx, W, swz)) = pp[u];
                        part += d.x * yy[u].x + d.y * yy[u].y + d.z * yy[u].z + d.w * yy[u].w;
                    }
                }
#pragma unroll
                for (int o = 8; o > 0; o >>= 1) part += __shfl_xor(part, o);
                if (sl == 0) rd_s[i] = part;
            }
        }
        __syncthreads();

        // ---- phase 1: dP[j, cols] = sum_i A^[i,j] dY'[i, cols] on the matrix cores ---------------------------------------------
        // unit = (16-row tile of the molecule, column group): the column tiles of a piece are split over the waves when the
        // molecule has fewer than four row tiles; at most three column tiles per unit (MB_MAXCW = 12)
        for (int u = wave; u < nunits; u += 4) {
            const int rt = u / CG, cg = u - rt * CG;
            const int ta = (cg * cw) / CG, nt = ((cg + 1) * cw) / CG - ta;
            const int ja = rt * 16 + li;
            f32x4 acc[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) acc[c] = (f32x4){0.f, 0.f, 0.f, 0.f};
            uint32_t cc[4];
            if (u == wave) {
#pragma unroll
                for (int e = 0; e < 4; ++e) cc[e] = cpre[e];
            } else {
                load_codes(rt, 0, cc);
            }
            for (int ib = 0; ib < n; ib += 64) {
                uint32_t cn[4] = {0u, 0u, 0u, 0u};
                if (ib + 64 < n) load_codes(rt, ib + 64, cn);          // next 64 source rows: in flight under this block's MFMAs
#pragma unroll 1
                for (int g4 = 0; g4 < 4; ++g4) {                        // 16 source rows = four k-steps at a time (rolled: the
                    if (ib + 16 * g4 >= n) break;                       // unrolled form kept 48 LDS operands in flight and spilled)
                    const uint32_t word = g4 == 0 ? cc[0] : (g4 == 1 ? cc[1] : (g4 == 2 ? cc[2] : cc[3]));
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        const int i0 = ib + 16 * g4 + 4 * t;            // first source row of this k-step
                        if (i0 < n) {                                    // wave-uniform
                            const int i = i0 + q;
                            const bool ok = i < n;
                            const float rs = (ok && ja < n) ? rs_s[i] : 0.0f;
                            const uint32_t ce = (word >> (8 * t)) & 255u;
                            float uu = sig_s[ce] + (ce == 0u ? TINY : 0.0f);
                            if (i == ja) uu += r;
                            uu *= rs;
#pragma unroll
                            for (int c = 0; c < 3; ++c)
                                if (c < nt) {
                                    const float bv = ok ? dYs[mol_at(i, (ta + c) * 16 + li, W, swz)] : 0.0f;
                                    acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(uu, bv, acc[c], 0, 0, 0);
                                }
                        }
                    }
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) cc[e] = cn[e];
            }
#pragma unroll
            for (int c = 0; c < 3; ++c)
                if (c < nt) {
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const int row = rt * 16 + q * 4 + g;
                        if (row < n) a.dP[(size_t)(r0 + row) * a.ld + c0 + (ta + c) * 16 + li] = acc[c][g];
                    }
                }
        }

        // ---- phase 2: edge gradients of this piece's columns (closed form in agg.hip / SURVEY.md 8a) --------------------------
        // one packed row per 16-lane group; the bonds found in the row's code bytes are queued in LDS and their dot products
        // with the P rows taken from the staged slice
        {
            double dr_acc = 0.0;
            for (int i = grp; i < RT * 16; i += 16) {                   // (wave-uniform trip count)
                const bool inrow = i < n;
                const float rs = inrow ? rs_s[i] : 0.0f;
                const bool live = rs != 0.0f;                           // m_i == 0 rows carry no dependence
                const int nn = live ? n : 0;
                const float rd = inrow ? rd_s[i] : 0.0f;
                float dyv[MB_MAXCW];
#pragma unroll
                for (int m = 0; m < MB_MAXCW; ++m) dyv[m] = (m < cw && inrow) ? dYs[mol_at(i, sl + 16 * m, W, swz)] : 0.0f;
                const uint8_t* crow = codeb + (size_t)(inrow ? i : 0) * bt.ldc;
                const int gq = grp;
                int nh = 0;
                auto flush = [&]() {
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    for (int h = 0; h < nh; ++h) {
                        const int hv = hit_s[gq][h];
                        const int j = hv >> 8;
                        float g = 0.0f;
#pragma unroll
                        for (int m = 0; m < MB_MAXCW; ++m)
                            if (m < cw) g += dyv[m] * Ps[mol_at(j, sl + 16 * m, W, swz)];
#pragma unroll
                        for (int o = 8; o > 0; o >>= 1) g += __shfl_xor(g, o);
                        if (sl == 0) {
                            const float dU = rs * (g - rd);
                            const uint32_t cj = (uint32_t)hv & 255u;
                            if (cj) {
                                const float sg = sig_s[cj];
                                atomicAdd(&h_s[cj], (double)(dU * sg * (1.0f - sg)));
                            }
                            if (j == i) dr_acc += (double)dU;
                        }
                    }
                    nh = 0;
                };
                for (int seg = 0; seg * 256 < n; ++seg) {
                    const int jb = seg * 256 + sl * 16;
                    uint4 cw4 = make_uint4(0u, 0u, 0u, 0u);
                    if (seg == 0 && i == grp) cw4 = crow_pre;
                    else if (jb < nn) cw4 = *reinterpret_cast<const uint4*>(crow + jb);
                    if (jb >= nn) cw4 = make_uint4(0u, 0u, 0u, 0u);
                    const uint32_t w[4] = {cw4.x, cw4.y, cw4.z, cw4.w};
#pragma unroll
                    for (int t = 0; t < 16; ++t) {
                        const uint32_t c = (w[t >> 2] >> (8 * (t & 3))) & 255u;
                        const int j = jb + t;
                        const bool want = j < nn && (c != 0u || j == i);
                        const unsigned long long ball = __ballot(want);
                        if (ball == 0ull) continue;                            // wave-uniform
                        const uint32_t mask = (uint32_t)(ball >> ((lane >> 4) * 16)) & 0xFFFFu;   // this group's lanes with a hit
                        if (mask) {
                            const int pos = nh + __popc(mask & ((1u << sl) - 1u));
                            if (want && pos < MB_HMAX) hit_s[gq][pos] = (j << 8) | (int)c;
                            nh = min(nh + __popc(mask), MB_HMAX);
                        }
                        if (nh > MB_HMAX - 16) flush();
                    }
                }
                flush();
            }
            if (sl == 0 && dr_acc != 0.0) atomicAdd(&dr_s[grp], dr_acc);
        }
        __syncthreads();
        // ---- bond-type histogram of this workgroup -> one of eight fp64 accumulator copies (only the non-zero bins) ---------
        {
            double* out = a.eacc + ((size_t)(blockIdx.x & 7) * a.vc.K + k) * EDGE_SLAB;
            const double v = h_s[tid];
            if (v != 0.0) atomicAdd(&out[tid], v);
            if (tid == 0) {
                double t = 0.0;
                for (int g = 0; g < 16; ++g) t += dr_s[g];
                if (t != 0.0) atomicAdd(&out[256], t);
            }
        }
    }
    // ---- ticket: the last workgroup drains the accumulators (atomic exchange: reads AND leaves zero) and writes the gradients.
    //      Only workgroups inside the device-side tile count take part (the grid is sized for the tile CAPACITY: tens of
    //      thousands of arrivals on a counter cost more than the kernel); an empty batch is finished by workgroup (0, 0). ------
    if ((int)blockIdx.x < ntiles) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // this wave's atomics have been performed
        __syncthreads();
        if (tid == 0) last_s = ticket_arrive(a.ticket, (int)blockIdx.x, ntiles, (int)gridDim.y) ? 1 : 0;
        __syncthreads();
        if (!last_s) return;
    } else if (!(ntiles == 0 && blockIdx.x == 0 && blockIdx.y == 0)) {
        return;
    }
    double* tot = reinterpret_cast<double*>(smem);              // [K][EDGE_SLAB] (the launcher sizes the dynamic LDS for it)
    const int K = a.vc.K;
    for (int e = tid; e < K * EDGE_SLAB; e += 256) {
        const int kk = e / EDGE_SLAB, c = e - kk * EDGE_SLAB;
        double t = 0.0;
        if (c <= 256) {
            unsigned long long v[8];
#pragma unroll
            for (int cp = 0; cp < 8; ++cp)
                v[cp] = atomicExch(reinterpret_cast<unsigned long long*>(a.eacc + ((size_t)cp * K + kk) * EDGE_SLAB + c), 0ull);
#pragma unroll
            for (int cp = 0; cp < 8; ++cp) t += __longlong_as_double((long long)v[cp]);
        }
        tot[e] = t;
    }
    __syncthreads();
    for (int e = tid; e < K * EDGE_SLAB; e += 256) {
        const int kk = e / EDGE_SLAB, c = e - kk * EDGE_SLAB;
        if (c == 256) {
            const double rr = (double)a.rsig[kk];
            a.dself_r[kk][0] = (float)(tot[e] * rr * (1.0 - rr));
        } else if (!a.rel_vec[kk]) {
            if (c >= 1 && c <= a.channels[kk]) a.datt_w[kk][c - 1] = (float)tot[e];
        } else if (c >= 1 && c <= a.rel_c[kk]) {
            // general relation vectors: the histogram is per bond CODE, the gradient per CHANNEL (layer.hip unpack_grads)
            const float* vec = a.rel_vec[kk];
            const int C = a.rel_c[kk], D = a.channels[kk];
            double t = 0.0;
            for (int code = 1; code <= D; ++code) t += tot[kk * EDGE_SLAB + code] * (double)vec[(size_t)(code - 1) * C + (c - 1)];
            a.datt_w[kk][c - 1] = (float)t;
        }
    }
}

// dynamic LDS of the kernel for a batch capacity N and at most `ctl` column tiles per y-chunk: the largest slice any molecule
// size n <= N can ask for (two [n][W] slices + two [n] vectors), and the [K][EDGE_SLAB] doubles of the final reduction
static size_t mol_bwd_lds_bytes(int N, int ctl, int K) {
    size_t fl = 0;
    for (int n = 1; n <= N; ++n) {
        const int nc = (n + 15) / 16;
        const int cw = (ctl + nc - 1) / nc;                    // largest share of a piece
        fl = std::max(fl, (size_t)2 * n * cw * 16 + 2 * n);
    }
    return std::max(fl * sizeof(float), (size_t)K * EDGE_SLAB * sizeof(double));
}

bool mol_bwd_enabled() {
    static const bool on = [] { const char* v = getenv("EAGCN_MOLBWD"); return !(v && v[0] == '0'); }();
    return on;
}

bool mol_bwd_ok(const eagcn_batch* b, const ViewCols& vc) {
    if (!mol_bwd_enabled() || b->n_tiles <= 0 || b->T <= 0) return false;
    int tmax = 0;
    for (int k = 0; k < vc.K; ++k) tmax = std::max(tmax, (vc.off[k + 1] - vc.off[k]) / 16);
    const int nchunk = cdiv(tmax, MB_MAXCW);
    const int CT = cdiv(tmax, nchunk);
    return mol_bwd_lds_bytes(b->N, CT, vc.K) <= 64 * 1024 && (long)vc.K * nchunk <= 65535;
}

int launch_mol_bwd(MolBwdArgs a, hipStream_t s) {
    int tmax = 0;
    for (int k = 0; k < a.vc.K; ++k) tmax = std::max(tmax, (a.vc.off[k + 1] - a.vc.off[k]) / 16);
    a.nchunk = cdiv(tmax, MB_MAXCW);
    a.CT = cdiv(tmax, a.nchunk);
    const size_t lds = mol_bwd_lds_bytes(a.bt.N, a.CT, a.vc.K);
    EAGCN_CHECK_ARG(lds <= 64 * 1024, "mol_bwd: %zu bytes of LDS for N=%d", lds, a.bt.N);
    dim3 grid(std::max(1, a.bt.n_tiles), a.vc.K * a.nchunk);
    ProfScope ps(PROF_AGG, s);
    mol_bwd_kernel<<<grid, 256, lds, s>>>(a);
    EAGCN_LAUNCH_CHECK();
    return EAGCN_OK;
}

}  // namespace eagcn
