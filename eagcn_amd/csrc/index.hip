// Batch index: turns the reference's dense collate tensors (utils.py:575-640) into the compact
// structures every other kernel consumes.
//
// What the reference does with these inputs, per layer and per view (layers.py:82-83, 294-304):
//   S = conv1x1(R_k) -> sigmoid -> * adj ;  mask = adj.max(dim=2) ; identity = mask * I
// A1 = sigmoid(S) * adj is zero wherever adj is zero, so R_k only matters at bonded (i,j): this
// kernel streams adj once (coalesced, HBM-bound, 4*N*N bytes per molecule) and gathers the C_k
// channel values only at the non-zeros of adj, producing one uint8 bond-type code per (i,j,view).
// With one-hot channels (neural_fp.py:111-120) the 1x1 conv is exactly a dictionary lookup
// sigma(w_k[type]); the gather validates one-hotness and reports violations in meta[] so the host
// side can fail loudly instead of computing something else.
#include <algorithm>

#include <stdlib.h>

#include "common.h"
#include "kernels.h"

namespace eagcn {

struct RelPtrs {
    const float* p[EAGCN_MAX_VIEWS];
    int c[EAGCN_MAX_VIEWS];
};

// one wavefront per RPW consecutive padded rows.  Every lane owns 4 consecutive columns per trip (NIT =
// ceil(ldc/256) trips); ALL adjacency loads of the RPW rows are issued up front (one row per wave kept a
// single 576-byte request in flight and ran at 1.1 TB/s on a 4096-molecule batch), and the codes of a view
// leave as one coalesced 32-bit store per lane (ldc is a multiple of 16).
template <int NIT, int RPW>
__global__ __launch_bounds__(256) void index_scan_kernel(const float* __restrict__ adj, RelPtrs rel,
                                                          int B, int N, int Ncap, int K, int ldc,
                                                          uint8_t* __restrict__ code,
                                                          int32_t* __restrict__ deg_bn,
                                                          int32_t* __restrict__ nat,
                                                          int32_t* __restrict__ ecnt,
                                                          int32_t* __restrict__ meta) {
    const int lane = threadIdx.x & 63;
    const long row_first = ((long)blockIdx.x * 4 + (threadIdx.x >> 6)) * RPW;      // b*N + i of the first row
    const long nrows = (long)B * N;
    if (row_first >= nrows) return;
    const size_t plane = (size_t)N * N;
    float a[RPW][NIT][4];
#pragma unroll
    for (int rr = 0; rr < RPW; ++rr) {
        const long row = row_first + rr;
        const float* arow = adj + (size_t)row * N;
#pragma unroll
        for (int t = 0; t < NIT; ++t)
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int j = (lane + 64 * t) * 4 + u;
                a[rr][t][u] = (row < nrows && j < N) ? arow[j] : 0.0f;
            }
    }
    int bad_adj = 0, bad_rel = 0;
#pragma unroll
    for (int rr = 0; rr < RPW; ++rr) {
        const long row = row_first + rr;
        if (row >= nrows) break;                                   // wave-uniform
        const int b = (int)(row / N), i = (int)(row % N);
        int deg = 0;
        uint32_t packed[NIT][EAGCN_MAX_VIEWS];
#pragma unroll
        for (int t = 0; t < NIT; ++t) {
            const int j0 = (lane + 64 * t) * 4;
#pragma unroll
            for (int k = 0; k < EAGCN_MAX_VIEWS; ++k) packed[t][k] = 0u;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float av = a[rr][t][u];
                const bool bond = (av != 0.0f);
                if (bond && av != 1.0f) ++bad_adj;
                deg += bond ? 1 : 0;
                if (bond) {
                    const int j = j0 + u;
#pragma unroll
                    for (int k = 0; k < EAGCN_MAX_VIEWS; ++k) {
                        if (k < K) {
                            const float* r = rel.p[k] + (size_t)b * rel.c[k] * plane + (size_t)i * N + j;
                            int ones = 0, other = 0, hot = 0;
                            // branch-free and unrolled: 8 independent strided loads in flight per lane
#pragma unroll 8
                            for (int ch = 0; ch < rel.c[k]; ++ch) {
                                const float v = r[(size_t)ch * plane];
                                const bool one = (v == 1.0f);
                                ones += one ? 1 : 0;
                                hot = one ? ch : hot;
                                other += (v != 0.0f && !one) ? 1 : 0;
                            }
                            if (ones != 1 || other != 0) ++bad_rel;
                            packed[t][k] |= (uint32_t)(hot + 1) << (8 * u);
                        }
                    }
                }
            }
        }
        deg = wave_sum(deg);
        // A row without bonds keeps whatever its code row held: every consumer weights such a row by
        // m_i / rowsum_i = 0 (agg.hip: rscale; the edge gradients skip it) and rows beyond nat[b] are never read,
        // so the 86 % padding rows of a Tox21-shaped batch cost one adjacency read and no K*ldc bytes of zero stores.
        if (deg > 0) {
#pragma unroll
            for (int t = 0; t < NIT; ++t) {
                const int j0 = (lane + 64 * t) * 4;
                if (j0 < ldc) {
#pragma unroll
                    for (int k = 0; k < EAGCN_MAX_VIEWS; ++k)
                        if (k < K)
                            *reinterpret_cast<uint32_t*>(code + (((size_t)k * B + b) * Ncap + i) * ldc + j0) = packed[t][k];
                }
            }
        }
        if (lane == 0) {
            deg_bn[(size_t)b * Ncap + i] = deg;       // (N = padded size of the caller's tensors, Ncap = capacity of the index)
            if (deg > 0) {                           // (no single-word counters here: 5k same-address atomics cost 60 us;
                atomicMax(&nat[b], i + 1);           //  these are one word per molecule)
                atomicAdd(&ecnt[b], deg);
            }
        }
    }
    bad_adj = wave_sum(bad_adj);
    bad_rel = wave_sum(bad_rel);
    if (lane == 0) {
        if (bad_adj) atomicAdd(&meta[EAGCN_META_BAD_ADJ], bad_adj);
        if (bad_rel) atomicAdd(&meta[EAGCN_META_BAD_REL], bad_rel);
    }
}

// ---- streaming scan (the default when N is a multiple of 4 and adj is 16-byte aligned) ----------------------------------------
// One wavefront takes RPW consecutive padded rows.  (1) ALL adjacency loads of those rows are issued up front as 16-byte loads
// (lane = four consecutive columns), so a wave has RPW x N x 4 bytes in flight instead of one row's 4-byte pieces.  (2) Degrees
// come from ballots (no cross-lane reduction).  (3) The bonds of the RPW rows are compacted into ONE list in LDS, and the
// relation channels of the whole list are gathered TOGETHER: lane c takes channel c of every bond, eight bonds per batch, so the
// ~38 single-sector loads per bond that one lane used to issue in five dependent batches are spread over the wave and overlap --
// one memory round trip per list of up to SCAN_CAP bonds instead of five per bonded row.  One-hotness is checked with one LDS
// word per (bond, view): +0x10000 + (channel + 1) for a channel equal to 1, +0x1000000 for any other non-zero value.
// (4) Every lane then packs the codes of its own four columns from that word and the code rows leave as before.
constexpr int SCAN_CAP = 64;                 // bonds gathered per pass (a wave whose rows hold more runs several passes)
// AL = elements per adjacency load: 4 (N a multiple of 4: every row 16-byte aligned), 2 (N even: 8-byte loads, e.g. the HIV
// set's N = 222) or 1 (odd N); the lane -> column map (lane f owns columns 4 f .. 4 f + 3) is the same in all three.
template <int NIT, int RPW, int AL>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3)))          // (168 registers; 190 and two waves per SIMD without the hint)
void index_scan4_kernel(const float* __restrict__ adj, RelPtrs rel, int B, int N, int Ncap, int K,
                                                           int ldc, uint8_t* __restrict__ code, int32_t* __restrict__ deg_bn,
                                                           int32_t* __restrict__ nat, int32_t* __restrict__ ecnt,
                                                           int32_t* __restrict__ meta) {
    __shared__ int list_s[4][SCAN_CAP];                  // per wave: (row in the wave << 10) | column
    __shared__ unsigned res_s[4][SCAN_CAP][EAGCN_MAX_VIEWS];
    __shared__ int rowb_s[4][RPW], rowi_s[4][RPW];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long row_first = ((long)blockIdx.x * 4 + wave) * RPW;
    const long nrows = (long)B * N;
    if (row_first >= nrows) return;                      // (wave-uniform)
    const size_t plane = (size_t)N * N;
    float4 a[RPW][NIT];
#pragma unroll
    for (int rr = 0; rr < RPW; ++rr) {
        const long row = row_first + rr;
#pragma unroll
        for (int t = 0; t < NIT; ++t) {
            const int j0 = (lane + 64 * t) * 4;
            a[rr][t] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (row < nrows && j0 < N) {
                const float* p = adj + (size_t)row * N + j0;
                if constexpr (AL == 4) a[rr][t] = *reinterpret_cast<const float4*>(p);
                else if constexpr (AL == 2) {
                    const float2 lo = *reinterpret_cast<const float2*>(p);
                    float2 hi = make_float2(0.f, 0.f);
                    if (j0 + 2 < N) hi = *reinterpret_cast<const float2*>(p + 2);
                    a[rr][t] = make_float4(lo.x, lo.y, hi.x, hi.y);
                } else {
                    a[rr][t].x = p[0];
                    if (j0 + 1 < N) a[rr][t].y = p[1];
                    if (j0 + 2 < N) a[rr][t].z = p[2];
                    if (j0 + 3 < N) a[rr][t].w = p[3];
                }
            }
        }
    }
    if (lane < RPW) {
        const long row = min(row_first + lane, nrows - 1);
        rowb_s[wave][lane] = (int)(row / N);
        rowi_s[wave][lane] = (int)(row % N);
    }
    // channel slot(s) of this lane: c = lane + 64 cc of the concatenated channel list -> (view, channel)
    int ctot = 0;
#pragma unroll
    for (int k = 0; k < EAGCN_MAX_VIEWS; ++k) ctot += k < K ? rel.c[k] : 0;
    const unsigned long long lt_mask = lane ? (~0ull >> (64 - lane)) : 0ull;
    // ---- degrees, bond positions ------------------------------------------------------------------------------------------
    int bad_adj = 0, bad_rel = 0;
    int deg[RPW];                                        // (wave-uniform)
    int posb[RPW][NIT];                                  // position of this lane's first bond of (row, trip) in the wave's list
    int total = 0;
#pragma unroll
    for (int rr = 0; rr < RPW; ++rr) {
        deg[rr] = 0;
#pragma unroll
        for (int t = 0; t < NIT; ++t) {
            const float v[4] = {a[rr][t].x, a[rr][t].y, a[rr][t].z, a[rr][t].w};
            int cnt = 0;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const bool bond = v[u] != 0.0f;
                if (bond && v[u] != 1.0f) ++bad_adj;
                cnt += bond ? 1 : 0;
            }
            // exclusive prefix of cnt (0..4) over the lanes: three ballots
            const unsigned long long b0 = __ballot(cnt & 1), b1 = __ballot(cnt & 2), b2 = __ballot(cnt & 4);
            const int before = __popcll(b0 & lt_mask) + 2 * __popcll(b1 & lt_mask) + 4 * __popcll(b2 & lt_mask);
            const int all = __popcll(b0) + 2 * __popcll(b1) + 4 * __popcll(b2);
            posb[rr][t] = total + before;
            total += all;
            deg[rr] += all;
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // ---- gather the relation channels of the bonds, SCAN_CAP at a time ------------------------------------------------------
    uint32_t packed[RPW][NIT][EAGCN_MAX_VIEWS];
#pragma unroll
    for (int rr = 0; rr < RPW; ++rr)
#pragma unroll
        for (int t = 0; t < NIT; ++t)
#pragma unroll
            for (int k = 0; k < EAGCN_MAX_VIEWS; ++k) packed[rr][t][k] = 0u;
    for (int base = 0; base < total; base += SCAN_CAP) {
        const int nb = min(SCAN_CAP, total - base);
        // (a) list + cleared result words of this pass
        for (int e = lane; e < nb; e += 64)
#pragma unroll
            for (int k = 0; k < EAGCN_MAX_VIEWS; ++k) res_s[wave][e][k] = 0u;
#pragma unroll
        for (int rr = 0; rr < RPW; ++rr)
#pragma unroll
            for (int t = 0; t < NIT; ++t) {
                const float v[4] = {a[rr][t].x, a[rr][t].y, a[rr][t].z, a[rr][t].w};
                int pos = posb[rr][t] - base;
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    if (v[u] != 0.0f) {
                        if (pos >= 0 && pos < nb) list_s[wave][pos] = (rr << 10) | ((lane + 64 * t) * 4 + u);
                        ++pos;
                    }
            }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        // (b) lane c: channel c of every listed bond, eight bonds in flight
        for (int c = lane; c < ctot; c += 64) {
            int k = 0, c0 = 0;                               // view whose channel range [c0, c0 + C_k) holds slot c
            {
                int start = 0;
#pragma unroll
                for (int v = 0; v < EAGCN_MAX_VIEWS; ++v)
                    if (v < K) {
                        if (c >= start) { k = v; c0 = start; }
                        start += rel.c[v];
                    }
            }
            const int ch = c - c0;
            const float* pk = nullptr;
            int ck = 1;
#pragma unroll
            for (int v = 0; v < EAGCN_MAX_VIEWS; ++v)
                if (v == k) { pk = rel.p[v]; ck = rel.c[v]; }
            for (int e0 = 0; e0 < nb; e0 += 8) {
                float val[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int e = min(e0 + u, nb - 1);
                    const int ent = list_s[wave][e];
                    const int rr = ent >> 10, j = ent & 1023;
                    val[u] = pk[((size_t)rowb_s[wave][rr] * ck + ch) * plane + (size_t)rowi_s[wave][rr] * N + j];
                }
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    if (e0 + u < nb && val[u] != 0.0f)
                        atomicAdd(&res_s[wave][e0 + u][k], val[u] == 1.0f ? (0x10000u + (unsigned)ch + 1u) : 0x1000000u);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        // (c) every lane: the codes of its own bonds of this pass
#pragma unroll
        for (int rr = 0; rr < RPW; ++rr)
#pragma unroll
            for (int t = 0; t < NIT; ++t) {
                const float v[4] = {a[rr][t].x, a[rr][t].y, a[rr][t].z, a[rr][t].w};
                int pos = posb[rr][t] - base;
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    if (v[u] != 0.0f) {
                        if (pos >= 0 && pos < nb) {
#pragma unroll
                            for (int k = 0; k < EAGCN_MAX_VIEWS; ++k)
                                if (k < K) {
                                    const unsigned w = res_s[wave][pos][k];
                                    const unsigned ones = (w >> 16) & 255u, other = w >> 24;
                                    if (ones != 1u || other != 0u) ++bad_rel;
                                    // (valid input: exactly one channel is 1 and the low half is its index + 1; otherwise the
                                    //  batch is rejected through meta[BAD_REL] and the value is irrelevant)
                                    packed[rr][t][k] |= ((ones == 1u ? (w & 0xFFFFu) : 1u) & 255u) << (8 * u);
                                }
                        }
                        ++pos;
                    }
            }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    // ---- code rows, degrees, per-molecule extents ---------------------------------------------------------------------------
#pragma unroll
    for (int rr = 0; rr < RPW; ++rr) {
        const long row = row_first + rr;
        if (row >= nrows) break;                         // wave-uniform
        const int b = rowb_s[wave][rr], i = rowi_s[wave][rr];
        // (a row without bonds keeps whatever its code row held: no consumer gives it weight, see index_scan_kernel)
        if (deg[rr] > 0) {
#pragma unroll
            for (int t = 0; t < NIT; ++t) {
                const int j0 = (lane + 64 * t) * 4;
                if (j0 < ldc) {
#pragma unroll
                    for (int k = 0; k < EAGCN_MAX_VIEWS; ++k)
                        if (k < K)
                            *reinterpret_cast<uint32_t*>(code + (((size_t)k * B + b) * Ncap + i) * ldc + j0) = packed[rr][t][k];
                }
            }
        }
        if (lane == 0) {
            deg_bn[(size_t)b * Ncap + i] = deg[rr];
            if (deg[rr] > 0) {
                atomicMax(&nat[b], i + 1);
                atomicAdd(&ecnt[b], deg[rr]);
            }
        }
    }
    bad_adj = wave_sum(bad_adj);
    bad_rel = wave_sum(bad_rel);
    if (lane == 0) {
        if (bad_adj) atomicAdd(&meta[EAGCN_META_BAD_ADJ], bad_adj);
        if (bad_rel) atomicAdd(&meta[EAGCN_META_BAD_REL], bad_rel);
    }
}

// compact input: one thread per directed bond writes its K codes and bumps the row degree.  Same contract as the dense
// scan: atoms inside the batch's padded size Nlog (<= the capacity N), a bond on the diagonal is a bond like any other (the
// dense signature accepts adj[i,i] = 1 too), and a directed bond may be listed only ONCE -- a repeated (b,i,j) would count
// its degree twice where the dense adjacency holds a single 1: the view-0 code byte is claimed with an atomic OR on its
// 32-bit word (the map is cleared before this kernel) and a second claim is reported in meta[BAD_ADJ].
__global__ __launch_bounds__(256) void index_bonds_kernel(const int32_t* __restrict__ bm, const int32_t* __restrict__ bi,
                                                           const int32_t* __restrict__ bj,
                                                           const uint8_t* __restrict__ bc, long E, int B, int N, int Nlog, int K,
                                                           int ldc, RelPtrs rel, uint8_t* __restrict__ code,
                                                           int32_t* __restrict__ deg_bn, int32_t* __restrict__ nat,
                                                           int32_t* __restrict__ ecnt, int32_t* __restrict__ meta) {
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < E; e += (long)gridDim.x * blockDim.x) {
        const int b = bm[e], i = bi[e], j = bj[e];
        if (b < 0 || b >= B || i < 0 || i >= Nlog || j < 0 || j >= Nlog) {
            atomicAdd(&meta[EAGCN_META_BAD_ADJ], 1);
            continue;
        }
        {
            const size_t at = (((size_t)b) * N + i) * ldc + j;             // view 0 (ldc is a multiple of 16: words never straddle rows)
            const unsigned sh = 8u * (unsigned)(at & 3);
            const unsigned c0 = (unsigned)min((int)bc[e * K], 254) + 1u;
            const unsigned old = atomicOr(reinterpret_cast<unsigned*>(code + (at & ~(size_t)3)), c0 << sh);
            if ((old >> sh) & 255u) {                                      // this (b,i,j) was listed before
                atomicAdd(&meta[EAGCN_META_BAD_ADJ], 1);
                continue;
            }
        }
        bool bad = bc[e * K] >= rel.c[0];
        for (int k = 1; k < K; ++k) {
            const int c = bc[e * K + k];
            if (c >= rel.c[k]) bad = true;
            code[(((size_t)k * B + b) * N + i) * ldc + j] = (uint8_t)(c + 1);
        }
        if (bad) atomicAdd(&meta[EAGCN_META_BAD_REL], 1);
        atomicAdd(&deg_bn[(size_t)b * N + i], 1);
        atomicMax(&nat[b], i + 1);
        atomicAdd(&ecnt[b], 1);
    }
}

// the small accumulators of an index build (meta, nat, ecnt; deg_bn when the scan does not visit every row) cleared by ONE launch:
// three or four hipMemsetAsync calls were seven runtime fill kernels of 5 us each at the head of the side stream's chain
struct ClearJob { int32_t* p[4]; int n[4]; };
__global__ __launch_bounds__(256) void index_clear_kernel(ClearJob j) {
#pragma unroll
    for (int r = 0; r < 4; ++r)
        for (int i = blockIdx.x * 256 + threadIdx.x; i < j.n[r]; i += gridDim.x * 256) j.p[r][i] = 0;
}
static int index_clear(int32_t* meta, int32_t* nat, int32_t* ecnt, int B, int32_t* deg_bn, long ndeg, hipStream_t s) {
    ClearJob j;
    j.p[0] = meta; j.n[0] = EAGCN_META_WORDS;
    j.p[1] = nat; j.n[1] = B;
    j.p[2] = ecnt; j.n[2] = ecnt ? B : 0;
    j.p[3] = deg_bn; j.n[3] = deg_bn ? (int)ndeg : 0;
    const long most = std::max<long>(B, j.n[3]);
    index_clear_kernel<<<(unsigned)std::min<long>(std::max<long>(1, (most + 255) / 256), 1024), 256, 0, s>>>(j);
    EAGCN_LAUNCH_CHECK();
    return EAGCN_OK;
}

// single workgroup: exclusive prefix sums of nat[] and ceil(nat/16) -> row0, tile0, totals
__global__ __launch_bounds__(1024) void index_offsets_kernel(const int32_t* __restrict__ nat, int B,
                                                              const int32_t* __restrict__ deg_bn, int BN,
                                                              int32_t* __restrict__ row0,
                                                              int32_t* __restrict__ tile0,
                                                              int32_t* __restrict__ meta, int cap_rows,
                                                              const int32_t* __restrict__ ecnt,
                                                              int32_t* __restrict__ edge0, int cap_edges, int n_logical) {
    __shared__ int s_rows[1024], s_tiles[1024], s_edges[1024];
    __shared__ int carry_r, carry_t, carry_e, s_max;
    const int t = threadIdx.x;
    if (t == 0) { carry_r = 0; carry_t = 0; carry_e = 0; s_max = 0; }
    __syncthreads();
    int nmax = 0;
    for (int base = 0; base < B; base += 1024) {
        int idx = base + t;
        int n = idx < B ? nat[idx] : 0;
        nmax = max(nmax, n);
        const int ne = (idx < B && ecnt) ? ecnt[idx] : 0;
        s_rows[t] = n;
        s_tiles[t] = (n + 15) >> 4;
        s_edges[t] = ne;
        __syncthreads();
        for (int o = 1; o < 1024; o <<= 1) {   // Hillis-Steele inclusive scan
            int vr = t >= o ? s_rows[t - o] : 0;
            int vt = t >= o ? s_tiles[t - o] : 0;
            int ve = t >= o ? s_edges[t - o] : 0;
            __syncthreads();
            s_rows[t] += vr;
            s_tiles[t] += vt;
            s_edges[t] += ve;
            __syncthreads();
        }
        if (idx < B) {
            row0[idx] = carry_r + s_rows[t] - n;
            tile0[idx] = carry_t + s_tiles[t] - ((n + 15) >> 4);
            if (edge0) edge0[idx] = carry_e + s_edges[t] - ne;
        }
        __syncthreads();
        if (t == 1023) { carry_r += s_rows[t]; carry_t += s_tiles[t]; carry_e += s_edges[t]; }
        __syncthreads();
    }
    atomicMax(&s_max, nmax);
    // directed bonds of the batch: the sum of the per-molecule counts (the prefix above already holds it); only a caller without
    // per-molecule counts pays for a pass over all B*N degrees (135 k entries through ONE workgroup: 110 us at B = 1024)
    if (!ecnt) {
        int edges = 0;
        for (int r0 = t; r0 < BN; r0 += 1024 * 8) {            // 8 loads in flight per thread
            int v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int r = r0 + 1024 * u;
                v[u] = r < BN ? deg_bn[r] : 0;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) edges += v[u];
        }
        edges = wave_sum(edges);
        if ((t & 63) == 0 && edges) atomicAdd(&meta[EAGCN_META_NEDGE], edges);
    }
    __syncthreads();
    if (t == 0) {
        row0[B] = carry_r;
        tile0[B] = carry_t;
        // a batch that does not fit the caller's row capacity is indexed as empty (and reported): every consumer
        // takes its extents from meta[], so nothing is read or written beyond the capacity-sized buffers
        if (edge0) edge0[B] = carry_e;
        if (ecnt) meta[EAGCN_META_NEDGE] = carry_e;
        const bool over_r = cap_rows > 0 && carry_r > cap_rows;
        const bool over_e = edge0 && carry_e > cap_edges;          // (the edge arrays are always capacity-sized)
        const bool over = over_r || over_e;
        meta[EAGCN_META_T] = over ? 0 : carry_r;
        meta[EAGCN_META_NTILES] = over ? 0 : carry_t;
        meta[EAGCN_META_OVERFLOW] = over_r ? carry_r : 0;
        meta[EAGCN_META_EDGE_OVERFLOW] = over_e ? carry_e : 0;
        meta[EAGCN_META_NMAX] = s_max;
        meta[EAGCN_META_NLOG] = n_logical;
    }
}

__global__ __launch_bounds__(256) void index_rows_kernel(eagcn_batch bt) {
    const int b = blockIdx.x;
    const int n = bt.nat[b], r0 = bt.row0[b], t0 = bt.tile0[b];
    for (int i = threadIdx.x; i < n && r0 + i < bt.T; i += blockDim.x) {       // (never beyond the row capacity)
        int d = bt.deg_bn[(size_t)b * bt.N + i];
        bt.row_mol[r0 + i] = b;
        bt.row_loc[r0 + i] = i;
        bt.row_deg[r0 + i] = d;
        bt.row_m[r0 + i] = d > 0 ? 1.0f : 0.0f;
        reinterpret_cast<int4*>(bt.row_info)[r0 + i] = make_int4(b, i, n, r0);
    }
    for (int t = threadIdx.x; t < (n + 15) / 16 && t0 + t < bt.n_tiles; t += blockDim.x) {
        bt.tile_mol[t0 + t] = b;
        reinterpret_cast<int4*>(bt.tile_info)[t0 + t] = make_int4(b, t, n, r0);
    }
    if (threadIdx.x == 0 && bt.mol_info) {
        const int e0 = bt.edge0[b];
        reinterpret_cast<int4*>(bt.mol_info)[b] = make_int4(r0 + n <= bt.T ? n : 0, r0, e0, bt.edge0[b + 1] - e0);
    }
}

// Row blocks of the LDS-staged aggregation (csrc/lagg.hip): whole molecules, packed greedily in batch order to at most LAGG_RB packed
// rows and LAGG_MAXM molecules per block (a molecule larger than LAGG_RB rows gets a block of its own: the launcher never takes such a
// batch).  Block q is described by two int4 records, blk[2q] = {first molecule, molecules, first packed row, rows} and blk[2q+1] =
// {first list entry, list entries, 0, 0} -- everything a workgroup needs to request its data with ONE dependent load.  The packing is
// sequential by nature; ONE wavefront does it 64 molecules at a time (prefix sum of the row counts, then one ballot per closed block).
// (runs as ONE extra workgroup of index_csr_kernel's grid -- its first wavefront: the two are independent, and a launch of its own was
//  10 us + a launch gap on the side stream's chain, which a step waits for when the host does not run far enough ahead)
__device__ __forceinline__ void index_blocks_body(const eagcn_batch& bt, const int rb, const int lane) {     // rb: rows a block may hold (<= LAGG_RB)
    const int T = bt.meta[EAGCN_META_T];
    int4* out = reinterpret_cast<int4*>(bt.blk);
    int nb = 0, start = 0, rows = 0, cnt = 0;                        // the open block: first molecule, rows and molecules so far
    int start_r0 = 0, start_e0 = 0;                                  // ... its first packed row and first list entry
    auto close = [&](int end, int end_e0) {                          // molecules [start, end)
        if (lane == 0) {
            out[2 * nb] = make_int4(start, end - start, start_r0, rows);
            out[2 * nb + 1] = make_int4(start_e0, end_e0 - start_e0, 0, 0);
        }
        ++nb; rows = 0; cnt = 0;
    };
    int last_e0 = 0;
    for (int base = 0; base < bt.B && T > 0; base += 64) {
        const int nv = min(64, bt.B - base);                         // molecules of this chunk
        // one batch of loads per chunk (a closed block used to cost three dependent loads: 0.77 ms for 1024 one-molecule blocks)
        const int n = lane < nv ? max(bt.nat[base + lane], 0) : 0;
        const int r0v = bt.row0[min(base + lane, bt.B)], e0v = bt.edge0[min(base + lane, bt.B)];
        const int e0_end = bt.edge0[min(base + 64, bt.B)];
        int pre = n;                                                 // inclusive prefix of the row counts
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(pre, o); if (lane >= o) pre += t; }
        int pos = 0, before = 0;                                     // first molecule of the chunk not yet placed, rows in front of it
        const int n_next = __shfl_down(n, 1), e0_next = __shfl_down(e0v, 1);
        while (pos < nv) {
            // a run of molecules none of which can share a block with its successor (batches of large molecules: every molecule its own
            // block) is closed by its lanes in parallel -- one closure per loop trip was 0.5 ms for 1024 molecules of 256 atoms
            if (cnt == 0) {
                const bool solo = lane >= pos && lane < nv - 1 && n > 0 && n + n_next > rb;
                const unsigned long long stop = ~__ballot(solo) & (~0ull << pos);
                const int run_end = min(stop ? __ffsll((long long)stop) - 1 : 64, nv - 1);
                if (run_end > pos) {
                    if (lane >= pos && lane < run_end) {
                        out[2 * (nb + lane - pos)] = make_int4(base + lane, 1, r0v, n);
                        out[2 * (nb + lane - pos) + 1] = make_int4(e0v, e0_next - e0v, 0, 0);
                    }
                    nb += run_end - pos;
                    pos = run_end;
                    before = __shfl(pre, pos - 1);
                    continue;
                }
            }
            // lanes pos .. fit into the open block as long as rows and molecule count allow (monotone in the lane); an EMPTY block
            // takes its first molecule whatever its size
            const bool fits = lane >= pos && lane < nv &&
                              ((rows + pre - before <= rb && cnt + lane - pos + 1 <= LAGG_MAXM) || (cnt == 0 && lane == pos));
            const int nfit = __popcll(__ballot(fits));
            if (nfit > 0) {
                if (cnt == 0) { start = base + pos; start_r0 = __shfl(r0v, pos); start_e0 = __shfl(e0v, pos); }
                const int upto = __shfl(pre, pos + nfit - 1);
                rows += upto - before; cnt += nfit; pos += nfit; before = upto;
            }
            if (pos < nv) close(base + pos, __shfl(e0v, pos));       // the next molecule does not fit: the block is complete
        }
        last_e0 = e0_end;
    }
    if (cnt > 0) close(bt.B, last_e0);
    if (lane == 0) bt.meta[EAGCN_META_NBLK] = nb;
}

// Bond lists of one molecule (one workgroup each) from its code maps: row lists (bonds (i,j) of row i, j ascending) and
// column lists (bonds (i,j) into column j, i ascending), each with the K per-view bond-type codes of the bond in one
// 64-bit word.  Only bonds with both ends inside the molecule's nat rows are listed (what the dense kernels multiply,
// agg.hip); rows without bonds have undefined code rows and are skipped.  Entries of molecule b live in
// [edge0[b], edge0[b+1]) of both list families.
// BITMAP: the bond positions of the molecule are first packed into an LDS bitmap ([nat][W] 32-bit words, one coalesced
// pass over the view-0 code rows), which both list families are then read from -- walking a COLUMN of the byte map in
// global memory is one dependent single-byte load per row (measured 256 us at the Tox21 shape).  Molecules beyond 512
// atom slots (bitmap > 32 KB) read the byte map directly.
template <bool BITMAP>
__global__ __launch_bounds__(256) void index_csr_kernel(eagcn_batch bt, int W, const int rb) {
    extern __shared__ uint32_t bits[];
    __shared__ int s_cnt[1024], s_ccnt[1024], s_scan[256];
    __shared__ unsigned char s_live[1024];
    if (rb > 0 && blockIdx.x == gridDim.x - 1) {                      // the extra workgroup: the row blocks of lagg.hip (index_blocks_body)
        if (threadIdx.x < 64) index_blocks_body(bt, rb, (int)threadIdx.x);
        return;
    }
    const int b = blockIdx.x, tid = threadIdx.x;
    const int n = bt.nat[b], r0 = bt.row0[b], e0 = bt.edge0[b];
    if (bt.meta[EAGCN_META_T] == 0 || n == 0 || r0 + n > bt.T) return;    // empty / overflowing batch: nothing is read
    const size_t plane = (size_t)bt.B * bt.N * bt.ldc;
    const uint8_t* code0 = bt.code + (size_t)b * bt.N * bt.ldc;          // view 0: every view has the same bond positions
    const int Wn = (n + 31) >> 5;
    auto bond = [&](int i, int j) -> bool {
        if constexpr (BITMAP) return (bits[i * W + (j >> 5)] >> (j & 31)) & 1u;
        else return s_live[i] && code0[(size_t)i * bt.ldc + j] != 0;
    };
    for (int i = tid; i < n; i += 256) {
        const bool lv = bt.deg_bn[(size_t)b * bt.N + i] > 0;
        s_live[i] = lv ? 1 : 0;
        int c = 0;
        for (int w = 0; w < Wn; ++w) {
            uint32_t m = 0u;
            if (lv) {
                const uint8_t* src = code0 + (size_t)i * bt.ldc + 32 * w;
                const uint4 a0 = *reinterpret_cast<const uint4*>(src);
                const uint4 a1 = (32 * w + 16 < bt.ldc) ? *reinterpret_cast<const uint4*>(src + 16) : make_uint4(0u, 0u, 0u, 0u);
                const uint32_t ww[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
#pragma unroll
                for (int t = 0; t < 32; ++t)
                    m |= (((ww[t >> 2] >> (8 * (t & 3))) & 255u) != 0u && 32 * w + t < n) ? (1u << t) : 0u;
            }
            if constexpr (BITMAP) bits[i * W + w] = m;
            c += __popc(m);
        }
        s_cnt[i] = c;
    }
    __syncthreads();
    // BITMAP: the transposed map (bond (i, j) -> bit i of row j) by LDS atomics, so that BOTH list families are walks over set bits
    // (first version: every thread scanned all n columns of its row and all n rows of its column, twice -- 1.0-2.0 ms per batch of
    // 256-atom molecules, 0.5 ms at the HIV shape, on the side stream but beside the step's own kernels)
    uint32_t* bitsT = bits + (BITMAP ? bt.N * W : 0);
    if constexpr (BITMAP) {
        for (int i = tid; i < n * W; i += 256) bitsT[i] = 0u;
        __syncthreads();
        for (int i = tid; i < n; i += 256)
            for (int w = 0; w < Wn; ++w) {
                uint32_t m = bits[i * W + w];
                while (m) {
                    const int j = 32 * w + __ffs(m) - 1;
                    m &= m - 1u;
                    atomicOr(&bitsT[j * W + (i >> 5)], 1u << (i & 31));
                }
            }
        __syncthreads();
    }
    for (int j = tid; j < n; j += 256) {                          // bonds into column j
        int cc = 0;
        if constexpr (BITMAP) {
            for (int w = 0; w < Wn; ++w) cc += __popc(bitsT[j * W + w]);
        } else {
            for (int i0 = 0; i0 < n; i0 += 8) {                   // 8 single-byte loads in flight
                uint8_t v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = (i0 + u < n && s_live[i0 + u]) ? code0[(size_t)(i0 + u) * bt.ldc + j] : 0;
#pragma unroll
                for (int u = 0; u < 8; ++u) cc += v[u] ? 1 : 0;
            }
        }
        s_ccnt[j] = cc;
    }
    __syncthreads();
    // exclusive scans of both count arrays (n <= 1024: four entries per thread, then a 256-entry scan)
    for (int pass = 0; pass < 2; ++pass) {
        int* cnt = pass ? s_ccnt : s_cnt;
        int v[4], tot = 0;
#pragma unroll
        for (int u = 0; u < 4; ++u) { const int i = tid * 4 + u; v[u] = i < n ? cnt[i] : 0; tot += v[u]; }
        s_scan[tid] = tot;
        __syncthreads();
        for (int o = 1; o < 256; o <<= 1) {
            const int x = tid >= o ? s_scan[tid - o] : 0;
            __syncthreads();
            s_scan[tid] += x;
            __syncthreads();
        }
        int run = s_scan[tid] - tot;
        __syncthreads();
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = tid * 4 + u;
            if (i < n) {
                int* ptr = (pass ? bt.col_ptr : bt.row_ptr) + (size_t)(r0 + i) * 2;
                const bool fits = e0 + run + v[u] <= bt.E;         // (always: index_offsets empties a batch that does not fit)
                ptr[0] = e0 + run;
                ptr[1] = fits ? v[u] : 0;
                cnt[i] = fits ? e0 + run : -1;
                run += v[u];
            }
        }
        __syncthreads();
    }
    auto codes = [&](int i, int j) -> uint64_t {                   // the K bond-type bytes of bond (i,j): K loads in flight
        uint8_t c[EAGCN_MAX_VIEWS];
#pragma unroll
        for (int k = 0; k < EAGCN_MAX_VIEWS; ++k) c[k] = k < bt.K ? code0[k * plane + (size_t)i * bt.ldc + j] : 0;
        uint64_t c64 = 0;
#pragma unroll
        for (int k = 0; k < EAGCN_MAX_VIEWS; ++k) c64 |= (uint64_t)c[k] << (8 * k);
        return c64;
    };
    for (int i = tid; i < n; i += 256) {
        int e = s_cnt[i];
        int ec = s_ccnt[i];                                      // column list of atom i: rows ascending
        if constexpr (BITMAP) {
            if (s_live[i] && e >= 0)
                for (int w = 0; w < Wn; ++w) {
                    uint32_t m = bits[i * W + w];
                    while (m) { const int j = 32 * w + __ffs(m) - 1; m &= m - 1u; bt.nbr[e] = j; bt.ecode[e] = codes(i, j); ++e; }
                }
            if (ec >= 0)
                for (int w = 0; w < Wn; ++w) {
                    uint32_t m = bitsT[i * W + w];
                    while (m) { const int ii = 32 * w + __ffs(m) - 1; m &= m - 1u; bt.tnbr[ec] = ii; bt.tcode[ec] = codes(ii, i); ++ec; }
                }
        } else {
            if (s_live[i] && e >= 0)
                for (int j = 0; j < n; ++j)
                    if (bond(i, j)) { bt.nbr[e] = j; bt.ecode[e] = codes(i, j); ++e; }
            if (ec >= 0)
                for (int ii = 0; ii < n; ++ii)
                    if (bond(ii, i)) { bt.tnbr[ec] = ii; bt.tcode[ec] = codes(ii, i); ++ec; }
        }
    }
}

// dense [B][N][F] -> packed [T][ld]; columns are re-grouped into the padded segments of `lay`
typedef ColMapD ColMap;
__device__ __forceinline__ int packed_to_exact(const ColMap& m, int cp) {
    int eo = 0, po = 0;
    for (int s = 0; s < m.nseg; ++s) {
        if (cp < po + m.p[s]) return (cp - po < m.w[s]) ? eo + (cp - po) : -1;
        eo += m.w[s];
        po += m.p[s];
    }
    return -1;
}

__global__ __launch_bounds__(256) void pack_rows_kernel(eagcn_batch bt, const float* __restrict__ dense,
                                                         int F, ColMap m, int ld, float* __restrict__ packed, int Nin) {
    const uint32_t total = (uint32_t)dev_rows(bt) * (uint32_t)ld;         // (rows x input features: far below 2^32)
    for (uint32_t e = blockIdx.x * blockDim.x + threadIdx.x; e < total; e += gridDim.x * blockDim.x) {
        int r = (int)(e / (uint32_t)ld), cp = (int)(e - (uint32_t)r * (uint32_t)ld);
        int ce = packed_to_exact(m, cp);
        float v = 0.0f;
        if (ce >= 0) v = dense[((size_t)bt.row_mol[r] * Nin + bt.row_loc[r]) * F + ce];      // Nin: padded size of the caller's tensor
        packed[e] = v;
    }
}

// unpadded per-molecule rows [sum n_b][F] (molecule b = rows off[b] .. off[b+1]) -> padded [B][N][F], zero beyond n_b:
// the device half of the reference's collate padding (utils.py:586-590 / 534-538)
__global__ __launch_bounds__(256) void pad_rows_kernel(const float* __restrict__ rows, const int32_t* __restrict__ off,
                                                        int B, int N, int F, float* __restrict__ out) {
    const size_t total = (size_t)B * N * F;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const int f = (int)(e % F);
        const size_t bi = e / F;
        const int i = (int)(bi % N), b = (int)(bi / N);
        const int o = off[b], n = off[b + 1] - o;
        out[e] = i < n ? rows[(size_t)(o + i) * F + f] : 0.0f;
    }
}

__global__ __launch_bounds__(256) void unpack_rows_kernel(eagcn_batch bt, const float* __restrict__ packed,
                                                           ColMap m, int ld, const float* __restrict__ pad_row,
                                                           float* __restrict__ dense, int F) {
    // one thread per dense element; exact column -> packed column
    const size_t total = (size_t)bt.B * bt.N * F;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total;
         e += (size_t)gridDim.x * blockDim.x) {
        int ce = (int)(e % F);
        size_t bi = e / F;
        int i = (int)(bi % bt.N), b = (int)(bi / bt.N);
        int eo = 0, po = 0, cp = -1;
        for (int s = 0; s < m.nseg; ++s) {
            if (ce < eo + m.w[s]) { cp = po + (ce - eo); break; }
            eo += m.w[s];
            po += m.p[s];
        }
        float v;
        if (i < bt.nat[b]) v = packed[(size_t)(bt.row0[b] + i) * ld + cp];
        else v = pad_row ? pad_row[cp] : 0.0f;
        dense[e] = v;
    }
}

}  // namespace eagcn

using namespace eagcn;

extern "C" int eagcn_index_build(const float* adj, const float* const* rel, eagcn_batch* b,
                                 int32_t* host_meta, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    EAGCN_CHECK_ARG(adj && rel && b && host_meta, "eagcn_index_build: null argument");
    EAGCN_CHECK_ARG(b->B > 0 && b->N > 0, "eagcn_index_build: B and N must be positive");
    EAGCN_CHECK_ARG(b->K >= 1 && b->K <= EAGCN_MAX_VIEWS, "eagcn_index_build: K=%d out of range", b->K);
    EAGCN_CHECK_ARG(b->ldc >= b->N && (b->ldc % 16) == 0, "eagcn_index_build: ldc must be a multiple of 16 >= N");
    EAGCN_CHECK_ARG(b->code && b->deg_bn && b->nat && b->row0 && b->tile0 && b->meta,
                    "eagcn_index_build: index buffers not allocated");
    RelPtrs rp;
    for (int k = 0; k < EAGCN_MAX_VIEWS; ++k) {
        rp.p[k] = k < b->K ? rel[k] : nullptr;
        rp.c[k] = k < b->K ? b->channels[k] : 0;
        if (k < b->K) {
            EAGCN_CHECK_ARG(rel[k] != nullptr, "eagcn_index_build: relation tensor %d is null", k);
            EAGCN_CHECK_ARG(rp.c[k] >= 1 && rp.c[k] <= EAGCN_MAX_CHANNELS,
                            "eagcn_index_build: view %d has %d channels (1..%d supported)", k, rp.c[k],
                            EAGCN_MAX_CHANNELS);
        }
    }
    const int Nin = b->n_logical > 0 ? b->n_logical : b->N;      // padded size (and row stride) of the caller's tensors
    EAGCN_CHECK_ARG(Nin <= b->N, "eagcn_index_build: n_logical %d exceeds the capacity N=%d", Nin, b->N);
    ProfScope ps(PROF_INDEX, s);
    EAGCN_CHECK_ARG(b->ecnt && b->edge0, "eagcn_index_build: bond-list buffers not allocated");
    EAGCN_CHECK_ARG((long)b->B * b->N < (1L << 31), "eagcn_index_build: B * N = %ld rows", (long)b->B * b->N);
    {   // (deg_bn: only the rows the scan does not visit)
        const int rcz = index_clear(b->meta, b->nat, b->ecnt, b->B, Nin < b->N ? b->deg_bn : nullptr, (long)b->B * b->N, s);
        if (rcz) return rcz;
    }
    const long rows = (long)b->B * Nin;
    // rows per wavefront: 4 was measured SLOWER (0.18 vs 0.09 ms at B=256, 1.12 vs 0.97 ms at B=4096), and so was
    // a streaming degree pass followed by a per-molecule code pass over the bonded rows (0.39-0.71 vs 0.07 ms at
    // B=256): the gather of one row (~38 single-sector loads from planes N*N floats apart per bond) takes tens of
    // microseconds however it is issued, so it has to run in as many waves at once as there are rows
    constexpr int RPW = 1;
    const unsigned sgrid = (unsigned)((rows + 4 * RPW - 1) / (4 * RPW));
    const int nit = cdiv((Nin + 15) / 16 * 16, 256);
    EAGCN_CHECK_ARG(nit <= 4, "eagcn_index_build: N=%d exceeds the supported 1024 atoms", b->N);
    // streaming scan: 16-byte adjacency loads, several rows per wave, the channel gather of a wave's bonds batched over its lanes
    static const bool scan4 = [] { const char* v = getenv("EAGCN_SCAN4"); return !(v && v[0] == '0'); }();
    if (scan4) {
        // widest aligned adjacency load the row stride and the base pointer allow
        const uintptr_t ap = reinterpret_cast<uintptr_t>(adj);
        const int al = ((Nin & 3) == 0 && (ap & 15) == 0) ? 4 : (((Nin & 1) == 0 && (ap & 7) == 0) ? 2 : 1);
#define EAGCN_SCAN4_AL(NIT, R, AL) index_scan4_kernel<NIT, R, AL><<<(unsigned)((rows + 4 * R - 1) / (4 * R)), 256, 0, s>>>(adj, rp, b->B, Nin, b->N, b->K, b->ldc, b->code, b->deg_bn, b->nat, b->ecnt, b->meta)
#define EAGCN_SCAN4(NIT, R) do { if (al == 4) EAGCN_SCAN4_AL(NIT, R, 4); else if (al == 2) EAGCN_SCAN4_AL(NIT, R, 2); else EAGCN_SCAN4_AL(NIT, R, 1); } while (0)
        if (Nin <= 256) EAGCN_SCAN4(1, 8);
        else if (Nin <= 512) EAGCN_SCAN4(2, 4);
        else EAGCN_SCAN4(4, 2);
#undef EAGCN_SCAN4
#undef EAGCN_SCAN4_AL
        EAGCN_LAUNCH_CHECK();
    } else
    {
#define EAGCN_SCAN(NIT) index_scan_kernel<NIT, RPW><<<sgrid, 256, 0, s>>>(adj, rp, b->B, Nin, b->N, b->K, b->ldc, b->code, b->deg_bn, b->nat, b->ecnt, b->meta)
        switch (nit) {
            case 1: EAGCN_SCAN(1); break;
            case 2: EAGCN_SCAN(2); break;
            case 3: EAGCN_SCAN(3); break;
            default: EAGCN_SCAN(4); break;
        }
#undef EAGCN_SCAN
        EAGCN_LAUNCH_CHECK();
    }
    index_offsets_kernel<<<1, 1024, 0, s>>>(b->nat, b->B, b->deg_bn, b->B * b->N, b->row0, b->tile0, b->meta, b->T, b->ecnt, b->edge0, b->E, b->n_logical > 0 ? b->n_logical : 0);
    EAGCN_LAUNCH_CHECK();
    EAGCN_HIP(hipMemcpyAsync(host_meta, b->meta, EAGCN_META_WORDS * sizeof(int32_t), hipMemcpyDeviceToHost, s));
    return EAGCN_OK;
}

extern "C" int eagcn_index_from_bonds(const int32_t* bond_mol, const int32_t* bond_i, const int32_t* bond_j,
                                      const uint8_t* bond_code, int64_t E, eagcn_batch* b, int32_t* host_meta,
                                      void* stream) {
    hipStream_t s = (hipStream_t)stream;
    EAGCN_CHECK_ARG(b && host_meta, "eagcn_index_from_bonds: null argument");
    EAGCN_CHECK_ARG(E == 0 || (bond_mol && bond_i && bond_j && bond_code), "eagcn_index_from_bonds: null bond arrays");
    EAGCN_CHECK_ARG(b->B > 0 && b->N > 0 && b->K >= 1 && b->K <= EAGCN_MAX_VIEWS, "eagcn_index_from_bonds: bad batch shape");
    EAGCN_CHECK_ARG(b->ldc >= b->N && (b->ldc % 16) == 0, "eagcn_index_from_bonds: ldc must be a multiple of 16 >= N");
    EAGCN_CHECK_ARG(b->code && b->deg_bn && b->nat && b->row0 && b->tile0 && b->meta,
                    "eagcn_index_from_bonds: index buffers not allocated");
    RelPtrs rp;
    for (int k = 0; k < EAGCN_MAX_VIEWS; ++k) { rp.p[k] = nullptr; rp.c[k] = k < b->K ? b->channels[k] : 0; }
    ProfScope ps(PROF_INDEX, s);
    EAGCN_CHECK_ARG(b->ecnt && b->edge0, "eagcn_index_from_bonds: bond-list buffers not allocated");
    EAGCN_CHECK_ARG((long)b->B * b->N < (1L << 31), "eagcn_index_from_bonds: B * N = %ld rows", (long)b->B * b->N);
    {
        const int rcz = index_clear(b->meta, b->nat, b->ecnt, b->B, b->deg_bn, (long)b->B * b->N, s);
        if (rcz) return rcz;
    }
    EAGCN_HIP(hipMemsetAsync(b->code, 0, (size_t)b->K * b->B * b->N * b->ldc, s));
    if (E > 0) {
        const int grid = (int)std::min<long>((E + 255) / 256, 4096);
        index_bonds_kernel<<<grid, 256, 0, s>>>(bond_mol, bond_i, bond_j, bond_code, (long)E, b->B, b->N,
                                                b->n_logical > 0 ? std::min(b->n_logical, b->N) : b->N, b->K, b->ldc, rp,
                                                b->code, b->deg_bn, b->nat, b->ecnt, b->meta);
        EAGCN_LAUNCH_CHECK();
    }
    index_offsets_kernel<<<1, 1024, 0, s>>>(b->nat, b->B, b->deg_bn, b->B * b->N, b->row0, b->tile0, b->meta, b->T, b->ecnt, b->edge0, b->E, b->n_logical > 0 ? b->n_logical : 0);
    EAGCN_LAUNCH_CHECK();
    EAGCN_HIP(hipMemcpyAsync(host_meta, b->meta, EAGCN_META_WORDS * sizeof(int32_t), hipMemcpyDeviceToHost, s));
    return EAGCN_OK;
}

extern "C" int eagcn_index_rows(const eagcn_batch* b, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    EAGCN_CHECK_ARG(b, "eagcn_index_rows: null batch");
    if (b->T == 0) return EAGCN_OK;
    EAGCN_CHECK_ARG(b->row_mol && b->row_loc && b->row_m && b->row_deg && b->tile_mol && b->row_info && b->tile_info,
                    "eagcn_index_rows: per-row buffers not allocated");
    EAGCN_CHECK_ARG(b->N <= 1024, "eagcn_index_rows: N=%d exceeds the supported 1024 atoms", b->N);
    EAGCN_CHECK_ARG(b->edge0 && b->mol_info && b->row_ptr && b->col_ptr && (b->E == 0 || (b->nbr && b->tnbr && b->ecode && b->tcode)),
                    "eagcn_index_rows: bond-list buffers not allocated");
    ProfScope ps(PROF_INDEX, s);
    index_rows_kernel<<<b->B, 256, 0, s>>>(*b);
    EAGCN_LAUNCH_CHECK();
    if (!b->build_lists) return EAGCN_OK;   // bond lists: GAT layers (gat.hip), bond-list aggregation (lagg.hip)
    const int rb = b->blk ? lagg_block_rows(b) : 0;                  // > 0: one more workgroup builds the row blocks
    const int W = (b->N + 31) / 32;
    // the transposed-bitmap form asks for 2 N W words of dynamic LDS on top of ~10 KB static: beyond 64 KB in total (N from ~465) the
    // launch needs the opt-in, and the device has to have that much (gfx950: 160 KB); otherwise the column scan
    const size_t dyn = (size_t)2 * b->N * W * sizeof(uint32_t);
    static const size_t lds_cap = [] {
        int dev = 0, v = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&v, hipDeviceAttributeMaxSharedMemoryPerBlock, dev) != hipSuccess) v = 64 * 1024;
        size_t cap = (size_t)v > 12 * 1024 ? (size_t)v - 12 * 1024 : 0;
        if (cap > 52 * 1024 &&
            hipFuncSetAttribute((const void*)index_csr_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)cap) != hipSuccess) {
            (void)hipGetLastError();
            cap = 52 * 1024;
        }
        return cap;
    }();
    if (b->N <= 512 && dyn <= lds_cap) index_csr_kernel<true><<<b->B + (rb > 0 ? 1 : 0), 256, dyn, s>>>(*b, W, rb);
    else index_csr_kernel<false><<<b->B + (rb > 0 ? 1 : 0), 256, 0, s>>>(*b, W, rb);
    EAGCN_LAUNCH_CHECK();
    return EAGCN_OK;
}

extern "C" int eagcn_pack_rows(const eagcn_batch* b, const float* dense, int F, const eagcn_layout* lay,
                               float* packed, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    EAGCN_CHECK_ARG(b && dense && lay && packed, "eagcn_pack_rows: null argument");
    EAGCN_CHECK_ARG(layout_width(lay) == F, "eagcn_pack_rows: layout width %d != F %d", layout_width(lay), F);
    if (b->T == 0) return EAGCN_OK;
    int ld = layout_ld(lay);
    size_t total = (size_t)b->T * ld;
    int grid = (int)std::min<size_t>((total + 255) / 256, 4096);
    ProfScope ps(PROF_PACK, s);
    pack_rows_kernel<<<grid, 256, 0, s>>>(*b, dense, F, make_colmap(lay), ld, packed, b->n_logical > 0 ? b->n_logical : b->N);
    EAGCN_LAUNCH_CHECK();
    return EAGCN_OK;
}

extern "C" int eagcn_pad_rows(const float* rows, const int32_t* mol_offset, int B, int N, int F, float* out, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    EAGCN_CHECK_ARG(rows && mol_offset && out && B > 0 && N > 0 && F > 0, "eagcn_pad_rows: bad argument");
    const size_t total = (size_t)B * N * F;
    ProfScope ps(PROF_PACK, s);
    pad_rows_kernel<<<(int)std::min<size_t>((total + 255) / 256, 4096), 256, 0, s>>>(rows, mol_offset, B, N, F, out);
    EAGCN_LAUNCH_CHECK();
    return EAGCN_OK;
}

extern "C" int eagcn_unpack_rows(const eagcn_batch* b, const float* packed, const eagcn_layout* lay,
                                 const float* pad_row, float* dense, int F, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    EAGCN_CHECK_ARG(b && lay && dense, "eagcn_unpack_rows: null argument");
    EAGCN_CHECK_ARG(b->T == 0 || packed, "eagcn_unpack_rows: null packed matrix");
    EAGCN_CHECK_ARG(layout_width(lay) == F, "eagcn_unpack_rows: layout width %d != F %d", layout_width(lay), F);
    int ld = layout_ld(lay);
    size_t total = (size_t)b->B * b->N * F;
    int grid = (int)std::min<size_t>((total + 255) / 256, 4096);
    ProfScope ps(PROF_PACK, s);
    unpack_rows_kernel<<<grid, 256, 0, s>>>(*b, packed, make_colmap(lay), ld, pad_row, dense, F);
    EAGCN_LAUNCH_CHECK();
    return EAGCN_OK;
}
