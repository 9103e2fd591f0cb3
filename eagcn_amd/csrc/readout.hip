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

// grid (B, ceil(F/64)); 4 wavefronts split the molecule's rows, 64 lanes own 64 exact columns
__global__ __launch_bounds__(256) void readout_fwd_kernel(eagcn_batch bt, const float* __restrict__ x, ColMapD m,
                                                           int ld, const float* __restrict__ pad_row,
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
        if (pad_row) s += (float)(bt.N - n) * pad_row[cp];
        const float inv = mode == 1 ? 1.0f / (float)size[b] : 1.0f;
        g[(size_t)b * F + f] = s * inv;
    }
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

// d pad_row[c] = sum_b (N - nat[b]) * dg[b][c] / size[b]
__global__ __launch_bounds__(256) void readout_bwd_pad_kernel(eagcn_batch bt, const float* __restrict__ dg, ColMapD m,
                                                               int ld, const int64_t* __restrict__ size, int mode,
                                                               int F, float* __restrict__ dpad) {
    const int cp = blockIdx.x * blockDim.x + threadIdx.x;
    if (cp >= ld) return;
    int eo = 0, po = 0, ce = -1;
    for (int s = 0; s < m.nseg; ++s) {
        if (cp < po + m.p[s]) { ce = (cp - po < m.w[s]) ? eo + (cp - po) : -1; break; }
        eo += m.w[s];
        po += m.p[s];
    }
    double acc = 0.0;
    if (ce >= 0)
        for (int b = 0; b < bt.B; ++b) {
            const float inv = mode == 1 ? 1.0f / (float)size[b] : 1.0f;
            acc += (double)((float)(bt.N - bt.nat[b]) * dg[(size_t)b * F + ce] * inv);
        }
    dpad[cp] = (float)acc;
}

int readout_backward_pad(const eagcn_batch* b, const float* dg, const eagcn_layout* lay, const int64_t* size,
                         int mode, int F, float* dpad_row, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    ProfScope ps(PROF_READOUT, s);
    readout_bwd_pad_kernel<<<cdiv(layout_ld(lay), 256), 256, 0, s>>>(*b, dg, make_colmap(lay), layout_ld(lay), size,
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
    readout_fwd_kernel<<<dim3(b->B, cdiv(F, 64)), 256, 0, s>>>(*b, x, make_colmap(lay), layout_ld(lay), pad_row, size, mode, g, F);
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
        readout_bwd_pad_kernel<<<cdiv(layout_ld(lay), 256), 256, 0, s>>>(*b, dg, make_colmap(lay), layout_ld(lay), size,
                                                                       mode, F, dpad_row);
        EAGCN_LAUNCH_CHECK();
    }
    return EAGCN_OK;
}
