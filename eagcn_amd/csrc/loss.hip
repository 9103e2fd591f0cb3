// Losses of the reference training loop, fused: value and gradient w.r.t. the logits in one launch.
//   classification (train.py:326-331 + utils.py:653-679): weight[b,t] = w[t][0] if label == 1,
//   w[t][1] if label == 0, else 0 (missing label); loss = sum_i weight_i * bce_with_logits(x_i, y_i)
//   / #labels in {0,1}.   regression (train.py:321-325): mean squared error.
// The reference builds the weight tensor with a B x T Python double loop (10 ms per 64x12 batch).
#include <algorithm>

#include "common.h"

namespace eagcn {

// single workgroup: B*T is at most a few 10^4 elements.  A thread keeps up to LOSS_CACHE of its elements (logit,
// label, both class weights) in registers: all loads of the kernel are ONE batch, and the gradient pass reuses
// them (a loop of dependent load pairs per element made this 8 us for 3072 elements).
constexpr int LOSS_CACHE = 4;
__global__ __launch_bounds__(1024) void bce_loss_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                         const float* __restrict__ w, int B, int T,
                                                         float* __restrict__ loss, float* __restrict__ dx) {
    __shared__ double s_sum[16];
    __shared__ int s_cnt[16];
    const int n = B * T;
    const int nthr = blockDim.x;
    float xc[LOSS_CACHE], yc[LOSS_CACHE], w1c[LOSS_CACHE], w0c[LOSS_CACHE];
#pragma unroll
    for (int u = 0; u < LOSS_CACHE; ++u) {
        const int i = threadIdx.x + u * nthr;
        const bool ok = i < n;
        const int t = ok ? i % T : 0;
        xc[u] = ok ? x[i] : 0.0f;
        yc[u] = ok ? y[i] : -1.0f;                     // -1 = "no label": weight 0, not counted
        w1c[u] = w[t * 2 + 0];
        w0c[u] = w[t * 2 + 1];
    }
    double acc = 0.0;
    int cnt = 0;
    auto term = [&](float xi, float yi, float w1, float w0) {
        const float wi = yi == 1.0f ? w1 : (yi == 0.0f ? w0 : 0.0f);
        cnt += (yi == 1.0f || yi == 0.0f) ? 1 : 0;
        // max(x,0) - x*y + log1p(exp(-|x|)): the numerically stable form ATen uses
        const float l = fmaxf(xi, 0.0f) - xi * yi + log1pf(expf(-fabsf(xi)));
        acc += (double)(wi * l);
    };
#pragma unroll
    for (int u = 0; u < LOSS_CACHE; ++u) term(xc[u], yc[u], w1c[u], w0c[u]);
    for (int i = threadIdx.x + LOSS_CACHE * nthr; i < n; i += nthr) {      // beyond the cache (large B*T)
        const int t = i % T;
        term(x[i], y[i], w[t * 2 + 0], w[t * 2 + 1]);
    }
    acc = wave_sum(acc);
    cnt = wave_sum(cnt);
    if ((threadIdx.x & 63) == 0) { s_sum[threadIdx.x >> 6] = acc; s_cnt[threadIdx.x >> 6] = cnt; }
    __syncthreads();
    double tot = 0.0;
    int c = 0;
    for (int k = 0; k < (int)(blockDim.x >> 6); ++k) { tot += s_sum[k]; c += s_cnt[k]; }
    const float inv = 1.0f / (float)c;
    if (threadIdx.x == 0) loss[0] = (float)(tot / (double)c);
    auto grad = [&](float xi, float yi, float w1, float w0) {
        const float wi = yi == 1.0f ? w1 : (yi == 0.0f ? w0 : 0.0f);
        return wi * (1.0f / (1.0f + expf(-xi)) - yi) * inv;
    };
#pragma unroll
    for (int u = 0; u < LOSS_CACHE; ++u) {
        const int i = threadIdx.x + u * nthr;
        if (i < n) dx[i] = grad(xc[u], yc[u], w1c[u], w0c[u]);
    }
    for (int i = threadIdx.x + LOSS_CACHE * nthr; i < n; i += nthr) {
        const int t = i % T;
        dx[i] = grad(x[i], y[i], w[t * 2 + 0], w[t * 2 + 1]);
    }
}

// Large batches (B*T beyond a few thousand): several workgroups.  d loss / d logit of an element needs the GLOBAL count of
// labelled entries, which every workgroup recounts for itself (labels are B*T floats, L2-resident); the loss value is the
// sum of the workgroups' partials (fp64 atomics on a device word that the last workgroup reads, publishes and re-arms).
// One call at a time per device (the accumulator is a module-level word): the training loop's single stream.
__device__ double g_bce_acc;
__device__ unsigned g_bce_ticket;
__global__ __launch_bounds__(1024) void bce_loss_multi_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                               const float* __restrict__ w, int B, int T,
                                                               float* __restrict__ loss, float* __restrict__ dx) {
    __shared__ double s_sum[16];
    __shared__ int s_cnt[16];
    const int n = B * T, nthr = blockDim.x;
    int cnt = 0;
    for (int i0 = threadIdx.x; i0 < n; i0 += nthr * 8) {               // 8 label loads in flight
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = i0 + u * nthr < n ? y[i0 + u * nthr] : -1.0f;
#pragma unroll
        for (int u = 0; u < 8; ++u) cnt += (v[u] == 1.0f || v[u] == 0.0f) ? 1 : 0;
    }
    cnt = wave_sum(cnt);
    if ((threadIdx.x & 63) == 0) s_cnt[threadIdx.x >> 6] = cnt;
    __syncthreads();
    int c = 0;
    for (int k = 0; k < (int)(blockDim.x >> 6); ++k) c += s_cnt[k];
    const float inv = 1.0f / (float)c;
    const int per = (n + gridDim.x - 1) / gridDim.x;
    const int lo = blockIdx.x * per, hi = min(n, lo + per);
    double acc = 0.0;
    for (int i = lo + threadIdx.x; i < hi; i += nthr) {
        const int t = i % T;
        const float xi = x[i], yi = y[i];
        const float wi = yi == 1.0f ? w[t * 2 + 0] : (yi == 0.0f ? w[t * 2 + 1] : 0.0f);
        acc += (double)(wi * (fmaxf(xi, 0.0f) - xi * yi + log1pf(expf(-fabsf(xi)))));
        dx[i] = wi * (1.0f / (1.0f + expf(-xi)) - yi) * inv;
    }
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) s_sum[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        double tot = 0.0;
        for (int k = 0; k < (int)(blockDim.x >> 6); ++k) tot += s_sum[k];
        atomicAdd(&g_bce_acc, tot);
        __threadfence();
        if (atomicAdd(&g_bce_ticket, 1u) == gridDim.x - 1) {           // the last workgroup: every partial has been added
            const double total = atomicAdd(&g_bce_acc, 0.0);
            loss[0] = (float)(total / (double)c);
            atomicExch((unsigned long long*)&g_bce_acc, 0ull);
            atomicExch(&g_bce_ticket, 0u);
        }
    }
}

__global__ __launch_bounds__(1024) void mse_loss_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                         int n, float* __restrict__ loss, float* __restrict__ dx) {
    __shared__ double s_sum[16];
    double acc = 0.0;
    const float inv = 1.0f / (float)n;
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const float d = x[i] - y[i];
        acc += (double)d * (double)d;
        dx[i] = 2.0f * d * inv;
    }
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) s_sum[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        double tot = 0.0;
        for (int k = 0; k < (int)(blockDim.x >> 6); ++k) tot += s_sum[k];
        loss[0] = (float)(tot / (double)n);
    }
}

// Evaluation outputs (train.py:130-211): the reference moves every batch's logits to the host, applies sigmoid there and
// walks B x T Python lists to drop the missing labels.  Here a batch is APPENDED to device-resident [cap, T] buffers:
// score = sigmoid(logit) (classification) or the prediction itself (regression), valid = label in {0, 1} (classification) or
// always 1; the metrics (AUC per task, RMSE) are then computed once over the buffers (eagcn_amd/training.py).
__global__ __launch_bounds__(256) void eval_append_kernel(const float* __restrict__ logits, const float* __restrict__ labels,
                                                           int n, int classification, float* __restrict__ scores,
                                                           float* __restrict__ targets, uint8_t* __restrict__ valid) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float x = logits[i], y = labels[i];
    scores[i] = classification ? 1.0f / (1.0f + expf(-x)) : x;
    targets[i] = y;
    valid[i] = classification ? ((y == 0.0f || y == 1.0f) ? 1 : 0) : 1;
}

}  // namespace eagcn

using namespace eagcn;

extern "C" int eagcn_eval_append(const float* logits, const float* labels, int B, int T, int classification, float* scores,
                                 float* targets, uint8_t* valid, int64_t row_offset, void* stream) {
    EAGCN_CHECK_ARG(logits && labels && scores && targets && valid, "eagcn_eval_append: null argument");
    EAGCN_CHECK_ARG(B > 0 && T > 0 && row_offset >= 0, "eagcn_eval_append: bad sizes");
    const int n = B * T;
    const size_t o = (size_t)row_offset * T;
    eval_append_kernel<<<cdiv(n, 256), 256, 0, (hipStream_t)stream>>>(logits, labels, n, classification, scores + o, targets + o, valid + o);
    EAGCN_LAUNCH_CHECK();
    return EAGCN_OK;
}

extern "C" int eagcn_bce_loss(const float* logits, const float* labels, const float* class_weight, int B, int T,
                              float* loss, float* dlogits, void* stream) {
    EAGCN_CHECK_ARG(logits && labels && class_weight && loss && dlogits, "eagcn_bce_loss: null argument");
    EAGCN_CHECK_ARG(B > 0 && T > 0, "eagcn_bce_loss: empty batch");
    const int n = B * T;
    if (n <= 4096) bce_loss_kernel<<<1, 1024, 0, (hipStream_t)stream>>>(logits, labels, class_weight, B, T, loss, dlogits);
    else bce_loss_multi_kernel<<<std::min(16, cdiv(n, 1024)), 1024, 0, (hipStream_t)stream>>>(logits, labels, class_weight, B, T, loss, dlogits);
    EAGCN_LAUNCH_CHECK();
    return EAGCN_OK;
}

extern "C" int eagcn_mse_loss(const float* pred, const float* target, int n, float* loss, float* dpred, void* stream) {
    EAGCN_CHECK_ARG(pred && target && loss && dpred, "eagcn_mse_loss: null argument");
    EAGCN_CHECK_ARG(n > 0, "eagcn_mse_loss: empty batch");
    mse_loss_kernel<<<1, 1024, 0, (hipStream_t)stream>>>(pred, target, n, loss, dpred);
    EAGCN_LAUNCH_CHECK();
    return EAGCN_OK;
}

// ---- optimizer step of the training loop (train.py:303 `optim.Adam(model.parameters(), lr, weight_decay=wd)`, train.py:334) -----------
// One launch over ONE flat fp32 buffer that holds every hot-path parameter (the gradients arrive in a buffer of the same layout:
// eagcn_amd/ops.py ModelPlan.offsets), instead of torch's foreach step over ~100 tensors.  torch.optim.Adam semantics, fp32:
//     g += wd p ; m = m + (1 - b1)(g - m) ; v = b2 v + (1 - b2) g g ; p -= lr / (1 - b1^t) * m / (sqrt(v) / sqrt(1 - b2^t) + eps)
// Capturable: the hyper-parameters and the step count live in DEVICE memory (a replayed graph re-reads them; a learning-rate
// schedule writes one float), the step count is advanced by the workgroup that finishes last (every workgroup has read it by
// then).  hyper = {lr, beta1, beta2, eps, weight_decay} as DOUBLES (torch's scalars are doubles).  ticket_dev == NULL: update this range,
// do not advance the count (FlatAdam's autograd mode updates the parameters that HAVE a gradient range by range and advances with the last).
namespace eagcn {
__global__ __launch_bounds__(256) void adam_step_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                        float* __restrict__ v, size_t n4, const double* __restrict__ hyper,
                                                        long long* step, unsigned* done) {
    // (every workgroup reads the count before the last one to finish advances it: a volatile read, `step` is not restrict)
    const long long t = *reinterpret_cast<volatile long long*>(step) + 1;
    // torch.optim.Adam forms every scalar in double precision on the host and hands the kernels fp32 images of them: 1 - beta2 from
    // the fp32 image of 0.999 is off by 1.7e-5 relative
    const double b1d = hyper[1], b2d = hyper[2];
    const float b1 = (float)b1d, b2 = (float)b2d, omb1 = (float)(1.0 - b1d), omb2 = (float)(1.0 - b2d);
    const float eps = (float)hyper[3], wd = (float)hyper[4];
    const double bc1 = 1.0 - pow(b1d, (double)t), bc2 = 1.0 - pow(b2d, (double)t);
    const float step_size = (float)(hyper[0] / bc1), inv_sqrt_bc2 = (float)(1.0 / sqrt(bc2));
    (void)b1;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        float4 pp = reinterpret_cast<float4*>(p)[i], mm = reinterpret_cast<float4*>(m)[i], vv = reinterpret_cast<float4*>(v)[i];
        const float4 gg = reinterpret_cast<const float4*>(g)[i];
        float* pe = &pp.x; float* me = &mm.x; float* ve = &vv.x; const float* ge = &gg.x;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float gr = ge[e] + wd * pe[e];
            me[e] = me[e] + omb1 * (gr - me[e]);
            ve[e] = b2 * ve[e] + omb2 * gr * gr;
            const float denom = sqrtf(ve[e]) * inv_sqrt_bc2 + eps;
            pe[e] = pe[e] - step_size * (me[e] / denom);
        }
        reinterpret_cast<float4*>(p)[i] = pp; reinterpret_cast<float4*>(m)[i] = mm; reinterpret_cast<float4*>(v)[i] = vv;
    }
    if (!done) return;                                 // (a partial update: the caller advances the count with its last range)
    __shared__ unsigned ticket;
    __syncthreads();
    if (threadIdx.x == 0) ticket = atomicAdd(done, 1u);
    __syncthreads();
    if (ticket == gridDim.x - 1 && threadIdx.x == 0) { *done = 0u; *step = t; }
}
}  // namespace eagcn

extern "C" int eagcn_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, const double* hyper_dev,
                               int64_t* step_dev, uint32_t* ticket_dev, void* stream) {
    EAGCN_CHECK_ARG(param && grad && exp_avg && exp_avg_sq && hyper_dev && step_dev, "eagcn_adam_step: null argument");
    EAGCN_CHECK_ARG(n >= 0 && (n & 3) == 0, "eagcn_adam_step: the flat buffers hold a multiple of 4 floats (ModelPlan pads every parameter)");
    EAGCN_CHECK_ARG(((reinterpret_cast<uintptr_t>(param) | reinterpret_cast<uintptr_t>(grad) | reinterpret_cast<uintptr_t>(exp_avg) |
                      reinterpret_cast<uintptr_t>(exp_avg_sq)) & 15) == 0, "eagcn_adam_step: buffers must be 16-byte aligned");
    if (n == 0) return EAGCN_OK;
    const size_t n4 = (size_t)n / 4;
    const unsigned grid = (unsigned)std::max<size_t>(1, std::min<size_t>((n4 + 255) / 256, 1024));
    eagcn::adam_step_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(param, grad, exp_avg, exp_avg_sq, n4, hyper_dev, (long long*)step_dev,
                                                                    ticket_dev);
    EAGCN_LAUNCH_CHECK();
    return EAGCN_OK;
}
