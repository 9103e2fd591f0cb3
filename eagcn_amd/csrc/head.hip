// Model-level entry points: the whole EAGCN forward (reference models.py:96-121) and its whole
// backward as ONE C call each, so the host issues a fixed, short launch sequence per step instead of
// walking an autograd graph of small ops.
//
//   forward : pack afm -> layer 1..L (layer.hip) -> read-out (models.py:108-111) -> Graph_BN ->
//             den1 -> bn_den1 -> relu -> dropout -> den2 (= graph_representation) -> bn_den2 -> relu
//             -> den3                                                      (models.py:112-120)
//   backward: the exact reverse, producing every parameter gradient.
//
// The head's dense layers and BatchNorm1d layers live in head2.hip: the producer of a matrix takes its
// column sums in the epilogue, the consumer normalises on the operand load, so no BatchNorm has a
// kernel of its own (except Graph_BN's backward, which has no dense layer in front of it).
#include <stdlib.h>

#include <algorithm>

#include "common.h"
#include "kernels.h"

namespace eagcn {

// step-position wait (eagcn_stream_wait_counter; the signal itself -- eagcn_model.fwd_signal -- is raised by the head's first launch, HeadFwd.signal)
// (gives up after `budget_ticks` of the 100 MHz clock and raises the sticky word `err`: a poll must never hang the queue)
__global__ void wait_counter_kernel(const uint32_t* __restrict__ word, uint32_t value, int* __restrict__ err,
                                    unsigned long long budget_ticks) {
    if (threadIdx.x == 0) {
        const unsigned long long t0 = wall_clock64();
        // (signed distance: the counter may wrap)
        while ((int32_t)(__hip_atomic_load(word, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) - value) < 0) {
            __builtin_amdgcn_s_sleep(64);
            if (wall_clock64() - t0 > budget_ticks) {
                if (err) __hip_atomic_store(err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                break;
            }
        }
    }
}



// batch-ready flag (eagcn_model.wait_flag, eagcn_stream_signal_flag): the stream that prepares a batch sets the word behind its
// last kernel, the step's first launch polls it and clears it.  Replaces hipStreamWaitEvent between the two streams, which costs
// the WAITING stream ~40 us per step on ROCm 7.2 even when the event completed long ago (tools/replay_probe.py: the captured
// configs[1] step replays in 355 us back to back, 395 us behind a cross-stream event wait).  A waiter that is never signalled
// (both streams on one hardware queue: the poll would block the signal) gives up after its budget and raises a sticky,
// host-visible word -- the caller checks it and falls back to events.
static int* g_wait_err_host = nullptr;
static int* g_wait_err_dev = nullptr;
static int* wait_err_word() {
    if (!g_wait_err_dev) {
        void* h = nullptr;
        if (hipHostMalloc(&h, 64, hipHostMallocMapped) != hipSuccess) return nullptr;
        memset(h, 0, 64);
        void* d = nullptr;
        if (hipHostGetDevicePointer(&d, h, 0) != hipSuccess) return nullptr;
        g_wait_err_host = (int*)h;
        g_wait_err_dev = (int*)d;
    }
    return g_wait_err_dev;
}
__global__ void signal_flag_kernel(uint32_t* __restrict__ flag) {
    if (threadIdx.x == 0) __hip_atomic_store(flag, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}
__global__ void wait_flag_kernel(uint32_t* __restrict__ flag, int* __restrict__ err, unsigned long long budget_ticks) {
    if (threadIdx.x == 0) {
        const unsigned long long t0 = wall_clock64();                   // 100 MHz
        while (__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) == 0u) {
            __builtin_amdgcn_s_sleep(8);
            if (wall_clock64() - t0 > budget_ticks) {
                if (err) __hip_atomic_store(err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                break;
            }
        }
        __hip_atomic_store(flag, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}
__global__ void handoff_kernel(HandOff h) {
    if (threadIdx.x == 0) handoff_body(h);
}
static int launch_wait_flag(uint32_t* flag, double budget_s, hipStream_t s) {
    wait_flag_kernel<<<1, 64, 0, s>>>(flag, wait_err_word(), (unsigned long long)(budget_s * 1e8));
    EAGCN_LAUNCH_CHECK();
    return EAGCN_OK;
}

// ---- carving of the saved-for-backward block and of the transient scratch ---------------------------
struct Carver2 {
    char* base;
    size_t off = 0;
    explicit Carver2(void* p) : base((char*)p) {}
    template <typename T>
    T* take(size_t n) {
        T* p = base ? (T*)(base + off) : nullptr;
        off = align256(off + std::max<size_t>(n, 1) * sizeof(T));
        return p;
    }
};

struct LayerSaved { float *P, *Y, *rscale, *bn, *xout, *pad_row; void* packed; size_t packed_bytes; int fp, ldo, ld_in;
                    uint16_t* xout_planes; };      // operand planes of the NEXT layer's products (gemm modes 3 / 4), or null
struct ModelSaved {
    float* x0;
    LayerSaved L[4];
    float *g, *h1, *h2, *bn_g, *bn_1, *bn_2;     // pre-BatchNorm matrices + [4][F] BN coefficient tables (the consumers re-normalise)
    uint16_t* pad_cnt;           // [B][K][ldo]: kept non-stored rows per (molecule, view, column) (readout.hip), or unused
    float* padc;                 // [B][ldo]
    uint32_t* pad_tab;           // [(N + 1)^2] binomial thresholds of the non-stored rows' kept counts (readout.hip), or unused
    size_t xout_last_off, pad_last_off;
};
// the last layer's non-stored rows go through a SAMPLED dropout when they reach the read-out unmasked
static bool pad_sampled(const eagcn_model* m) {
    const eagcn_layer_params* last = &m->layer[m->n_layers - 1];
    return last->structure == EAGCN_STRUCT_WEIGHTED && m->training && last->dropout > 0.0f;
}

static size_t carve_saved(void* base, const eagcn_batch* b, const eagcn_model* m, ModelSaved* out) {
    Carver2 c(base);
    ModelSaved s;
    const size_t T = (size_t)b->T;
    s.x0 = c.take<float>(T * layout_ld(&m->layer[0].in));
    for (int l = 0; l < m->n_layers; ++l) {
        const eagcn_layer_params* p = &m->layer[l];
        LayerSaved& L = s.L[l];
        L.fp = eagcn_layer_fp(p);
        L.ldo = eagcn_layer_out_ld(p);
        L.ld_in = layout_ld(&p->in);
        L.P = c.take<float>(T * L.fp);
        L.Y = c.take<float>(T * L.fp);
        L.rscale = c.take<float>((size_t)p->K * T);
        L.bn = c.take<float>((size_t)4 * L.fp);
        if (l == m->n_layers - 1) s.xout_last_off = c.off;
        L.xout = c.take<float>(T * L.ldo);
        if (l == m->n_layers - 1) s.pad_last_off = c.off;
        L.pad_row = c.take<float>(L.ldo);
        L.packed_bytes = eagcn_layer_packed_bytes(b, p);
        L.packed = c.take<char>(L.packed_bytes);
        // the layer above reads this layer's output through bf16 operand planes when its products run on gemm_bx3.hip
        const int np = (l + 1 < m->n_layers && L.ldo >= 128 && (L.ldo & 15) == 0) ? gemm_planes() : 0;
        L.xout_planes = np ? c.take<uint16_t>((size_t)np * bx_plane_elems(T, L.ldo)) : nullptr;
    }
    const eagcn_head_params* h = &m->head;
    const size_t B = (size_t)b->B;
    s.g = c.take<float>(B * h->f_in);
    s.h1 = c.take<float>(B * h->n_den1);
    s.h2 = c.take<float>(B * h->n_den2);
    s.bn_g = c.take<float>((size_t)4 * h->f_in);
    s.bn_1 = c.take<float>((size_t)4 * h->n_den1);
    s.bn_2 = c.take<float>((size_t)4 * h->n_den2);
    {
        const eagcn_layer_params* last = &m->layer[m->n_layers - 1];
        const size_t ldo = (size_t)eagcn_layer_out_ld(last);
        const bool need = last->structure == EAGCN_STRUCT_WEIGHTED;
        s.pad_cnt = c.take<uint16_t>(need ? B * last->K * ldo : 1);
        s.padc = c.take<float>(need ? B * ldo : 1);
        const size_t nt = (size_t)(b->N + 1) * (b->N + 1);
        s.pad_tab = c.take<uint32_t>(need ? nt : 1);
    }
    if (out) *out = s;
    return c.off;
}

struct ModelScratch {
    float *da2, *da1, *dgn, *dg, *dxa, *dxb, *dpad;
    float* hdw; int hdw_ks;      // row-chunk partials of the head's weight gradients (HeadBwd.ks): [ks][F n1 + n1 n2 + n2 nclass]
    double* hst; int n_hst;      // BatchNorm sums of the head: [f_in + n_den1 + n_den2][2], forward ...
    double* hws;                 // HEAD_WS doubles behind them (same cleared block): barrier words / label count / loss sum of head_all
    double* hsb;                 // ... and backward.  Both are cleared by the forward's parameter-packing launch; the backward clears
                                 // its own again on the way out (bn_bwd_reduce of the top layer), for a second backward call
    void* layer; size_t layer_bytes;
};
static size_t carve_scratch(void* base, const eagcn_batch* b, const eagcn_model* m, ModelScratch* out) {
    Carver2 c(base);
    ModelScratch s;
    const eagcn_head_params* h = &m->head;
    const size_t B = (size_t)b->B, T = (size_t)b->T;
    // three segments [2 w sums | row count | pad], one per BatchNorm of the head (the count rides with the sums through the
    // sync-BatchNorm hook)
    s.n_hst = 2 * (h->f_in + h->n_den1 + h->n_den2) + 6;
    s.hst = c.take<double>((size_t)(HEAD_COPIES + 1) * s.n_hst + HEAD_WS);     // HEAD_COPIES replicas of the forward sums (kernels.h)
    s.hsb = s.hst + (size_t)HEAD_COPIES * s.n_hst;
    s.hws = s.hsb + s.n_hst;
    s.da2 = c.take<float>(B * h->n_den2);
    s.da1 = c.take<float>(B * h->n_den1);
    s.dgn = c.take<float>(B * h->f_in);
    s.dg = c.take<float>(B * h->f_in);
    s.hdw_ks = head_dw_chunks(b->B);
    s.hdw = c.take<float>(s.hdw_ks > 1 ? (size_t)s.hdw_ks * ((size_t)h->f_in * h->n_den1 + (size_t)h->n_den1 * h->n_den2 +
                                                             (size_t)h->n_den2 * h->nclass) : 1);
    int ldmax = 0;
    size_t lbytes = 0;
    for (int l = 0; l < m->n_layers; ++l) {
        ldmax = std::max(ldmax, eagcn_layer_out_ld(&m->layer[l]));
        lbytes = std::max(lbytes, eagcn_layer_fwd_scratch_bytes(b, &m->layer[l]));
        lbytes = std::max(lbytes, eagcn_layer_bwd_scratch_bytes(b, &m->layer[l]));
    }
    s.dxa = c.take<float>(T * ldmax);
    s.dxb = c.take<float>(T * ldmax);
    s.dpad = c.take<float>((size_t)EAGCN_MAX_VIEWS * ldmax);
    s.layer_bytes = lbytes;
    s.layer = c.take<char>(lbytes);
    if (out) *out = s;
    return c.off;
}

static int check_model(const eagcn_batch* b, const eagcn_model* m, const char* who) {
    EAGCN_CHECK_ARG(b && m, "%s: null argument", who);
    EAGCN_CHECK_ARG(m->n_layers >= 1 && m->n_layers <= 4, "%s: n_layers=%d", who, m->n_layers);
    const eagcn_head_params* h = &m->head;
    EAGCN_CHECK_ARG(h->f_in >= 1 && h->n_den1 >= 1 && h->n_den2 >= 1 && h->nclass >= 1, "%s: bad head sizes", who);
    EAGCN_CHECK_ARG(h->den1_w && h->den2_w && h->den3_w && h->gbn_w && h->gbn_b && h->gbn_rm && h->gbn_rv &&
                        h->bn1_w && h->bn1_b && h->bn1_rm && h->bn1_rv && h->bn2_w && h->bn2_b && h->bn2_rm &&
                        h->bn2_rv, "%s: null head parameter", who);
    const eagcn_layer_params* last = &m->layer[m->n_layers - 1];
    const int f_last = last->structure == EAGCN_STRUCT_CONCATE ? [&] { int s = 0; for (int k = 0; k < last->K; ++k) s += last->width[k]; return s; }()
                                                                 : last->width[0];
    EAGCN_CHECK_ARG(f_last == h->f_in, "%s: head expects %d features, last layer yields %d", who, h->f_in, f_last);
    EAGCN_CHECK_ARG(!m->training || b->B > 1, "%s: BatchNorm in training mode needs more than one molecule", who);
    return EAGCN_OK;
}

static eagcn_layout out_layout(const eagcn_layer_params* p) {
    eagcn_layout l;
    memset(&l, 0, sizeof(l));
    if (p->structure == EAGCN_STRUCT_CONCATE) {
        l.nseg = p->K;
        for (int k = 0; k < p->K; ++k) { l.width[k] = p->width[k]; l.pad[k] = pad16(p->width[k]); }
    } else {
        l.nseg = 1;
        l.width[0] = p->width[0];
        l.pad[0] = pad16(p->width[0]);
    }
    return l;
}

// the top layer's output matrix is not built in the forward (eagcn_model.fuse_readout): Concate only -- its non-stored rows are
// masked to zero, so the read-out sums nothing but packed rows
static bool fused_readout(const eagcn_model* m) {
    static const bool env = [] { const char* v = getenv("EAGCN_NO_FUSED_READOUT"); return !(v && v[0] == '1'); }();
    return env && m->fuse_readout && m->layer[m->n_layers - 1].structure == EAGCN_STRUCT_CONCATE;
}

// layer l's output is needed as plane images only: it has them, and the layer above reads nothing else (forward and backward)
static bool hidden_planes_only(const eagcn_batch* b, const eagcn_model* m, const ModelSaved& sv, int l) {
    return l + 1 < m->n_layers && sv.L[l].xout_planes && layer_reads_planes_only(b, &m->layer[l + 1], m->aux_stream != nullptr);
}

static void fill_drop(float p, int training, int* do_drop, uint32_t* thr, float* inv_keep) {
    *do_drop = (training && p > 0.0f) ? 1 : 0;
    *thr = (uint32_t)std::min(4294967295.0, (double)p * 4294967296.0);
    *inv_keep = 1.0f / (1.0f - p);
}

}  // namespace eagcn

using namespace eagcn;

#define RC(call) do { int rc_ = (call); if (rc_) return rc_; } while (0)

extern "C" size_t eagcn_model_saved_bytes(const eagcn_batch* b, const eagcn_model* m) {
    return carve_saved(nullptr, b, m, nullptr);
}
extern "C" size_t eagcn_model_scratch_bytes(const eagcn_batch* b, const eagcn_model* m) {
    return carve_scratch(nullptr, b, m, nullptr);
}
extern "C" int eagcn_model_atom_rep(const eagcn_batch* b, const eagcn_model* m, size_t* xout_offset,
                                    size_t* pad_row_offset, int* ld) {
    EAGCN_CHECK_ARG(b && m && xout_offset && pad_row_offset && ld, "eagcn_model_atom_rep: null argument");
    ModelSaved s;
    carve_saved(nullptr, b, m, &s);
    *xout_offset = s.xout_last_off;
    *pad_row_offset = s.pad_last_off;
    *ld = eagcn_layer_out_ld(&m->layer[m->n_layers - 1]);
    return EAGCN_OK;
}

extern "C" int eagcn_model_atom_rep_materialize(const eagcn_batch* b, const eagcn_model* m, void* saved, size_t saved_bytes,
                                                 void* stream) {
    EAGCN_CHECK_ARG(b && m && saved, "eagcn_model_atom_rep_materialize: null argument");
    if (!fused_readout(m)) return EAGCN_OK;                  // the forward built the matrix itself
    ModelSaved sv;
    EAGCN_CHECK_ARG(carve_saved(saved, b, m, &sv) <= saved_bytes, "eagcn_model_atom_rep_materialize: saved block too small");
    const int l = m->n_layers - 1;
    LayerSaved& L = sv.L[l];
    eagcn_layer_bufs w;
    memset(&w, 0, sizeof(w));
    w.Y = L.Y; w.bn = L.bn; w.xout = L.xout; w.pad_row = L.pad_row; w.packed = L.packed; w.packed_bytes = L.packed_bytes;
    return layer_apply_impl(b, &m->layer[l], &w, stream);
}

extern "C" int eagcn_model_pack_input(const eagcn_batch* b, const eagcn_model* m, const float* afm, void* saved,
                                      size_t saved_bytes, void* stream) {
    EAGCN_CHECK_ARG(b && m && afm && saved, "eagcn_model_pack_input: null argument");
    ModelSaved sv;
    EAGCN_CHECK_ARG(carve_saved(saved, b, m, &sv) <= saved_bytes, "eagcn_model_pack_input: saved block too small");
    return eagcn_pack_rows(b, afm, layout_width(&m->layer[0].in), &m->layer[0].in, sv.x0, stream);
}

// descriptors of the head's stages (head2.hip), shared by the separate launches and by the one-launch training head
struct HeadPlan {
    HeadFwd f1, f2, f3;
    HeadBwd b3, b2, b1;
    HeadGbn bg;
    double *st_g, *st_1, *st_2, *sb_g, *sb_1, *sb_2;
    bool sync;
};
static HeadPlan head_plan(const eagcn_batch* b, const eagcn_model* m, const ModelSaved& sv, const ModelScratch& sc, float* out,
                          float* graph_rep, const float* dout, const float* dgraph_rep, const eagcn_head_grads* hg) {
    HeadPlan P;
    const eagcn_head_params* h = &m->head;
    const int B = b->B, F = h->f_in, n1 = h->n_den1, n2 = h->n_den2, nc = h->nclass;
    P.st_g = sc.hst; P.st_1 = sc.hst + 2 * F + 2; P.st_2 = sc.hst + 2 * (F + n1) + 4;
    P.sb_g = sc.hsb; P.sb_1 = sc.hsb + 2 * F + 2; P.sb_2 = sc.hsb + 2 * (F + n1) + 4;
    P.sync = m->stats_hook && m->training;
    // replicas of the forward sums (one with sync-BatchNorm: the hook all-reduces the first block in place)
    const int copies = P.sync ? 1 : HEAD_COPIES;
    HeadDrop nodrop{0, 0u, 1.0f, 0, nullptr}, drop1 = nodrop;
    {
        int on; uint32_t thr; float inv_keep;
        fill_drop(h->dropout, m->training, &on, &thr, &inv_keep);
        drop1 = HeadDrop{on, thr, inv_keep, m->head_seed, m->head_seed_dev};
    }
    P.f1 = HeadFwd{B, F, n1, sv.g, P.st_g, h->gbn_w, h->gbn_b, h->gbn_rm, h->gbn_rv, sv.bn_g, h->den1_w, sv.h1, nullptr, P.st_1,
                   m->training, 0, h->bn_eps, h->bn_momentum, nodrop};
    P.f2 = HeadFwd{B, n1, n2, sv.h1, P.st_1, h->bn1_w, h->bn1_b, h->bn1_rm, h->bn1_rv, sv.bn_1, h->den2_w, sv.h2, graph_rep, P.st_2,
                   m->training, 1, h->bn_eps, h->bn_momentum, drop1};
    P.f3 = HeadFwd{B, n2, nc, sv.h2, P.st_2, h->bn2_w, h->bn2_b, h->bn2_rm, h->bn2_rv, sv.bn_2, h->den3_w, out, nullptr, nullptr,
                   m->training, 1, h->bn_eps, h->bn_momentum, nodrop};
    if (P.sync) { P.f1.cnt_in = P.st_g + 2 * F; P.f2.cnt_in = P.st_1 + 2 * n1; P.f3.cnt_in = P.st_2 + 2 * n2; }
    P.f3.nw = 8;                         // (dense 3 adds in the order of the fused middle launch, whatever the batch size)
    P.f1.signal = m->fwd_signal;         // (the forward has passed its read-out: said by the head's first launch)
    P.f1.st_copies = P.f2.st_copies = P.f3.st_copies = copies;
    P.f1.st_stride = P.f2.st_stride = P.f3.st_stride = sc.n_hst;
    if (!hg) return P;
    // backward: den3 -> bn_den2 -> den2 -> bn_den1 -> den1 -> Graph_BN, one stage per dense layer (d input + d weight), each
    // BatchNorm's backward sums taken by the stage in front of it.  sync-BatchNorm: the global row counts are the ones the forward
    // left behind its statistics (same scratch block)
    const double *cn_g = sc.hst + 2 * F, *cn_1 = sc.hst + 2 * F + 2 + 2 * n1, *cn_2 = sc.hst + 2 * (F + n1) + 4 + 2 * n2;
    const float gscale = P.sync && m->stats_world > 1 ? 1.0f / (float)m->stats_world : 1.0f;
    // dense 3: y = out (no BatchNorm behind it), input a2 = relu(bn_den2(h2))
    P.b3 = HeadBwd{B, n2, nc, sv.h2, sv.bn_2, 1, nodrop, h->den3_w, dout, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr,
                   sc.da2, P.sb_2, hg->d_den3_w, m->training};
    // dense 2: y = h2 followed by bn_den2 (+ the gradient that reaches graph_representation directly), input a1
    P.b2 = HeadBwd{B, n1, n2, sv.h1, sv.bn_1, 1, drop1, h->den2_w, sc.da2, sv.h2, sv.bn_2, P.sb_2, dgraph_rep, hg->d_bn2_w, hg->d_bn2_b,
                   sc.da1, P.sb_1, hg->d_den2_w, m->training};
    // dense 1: y = h1 followed by bn_den1, input gn = Graph_BN(g)
    P.b1 = HeadBwd{B, F, n1, sv.g, sv.bn_g, 0, nodrop, h->den1_w, sc.da1, sv.h1, sv.bn_1, P.sb_1, nullptr, hg->d_bn1_w, hg->d_bn1_b,
                   sc.dgn, P.sb_g, hg->d_den1_w, m->training};
    if (P.sync) { P.b2.cnt_y = cn_2; P.b2.gscale = gscale; P.b1.cnt_y = cn_1; P.b1.gscale = gscale; }
    P.b3.nw = 8;
    // large batches: the weight gradients leave as row-chunk partials, summed with the Graph_BN backward
    const int ks = sc.hdw_ks;
    float* part3 = sc.hdw;
    float* part2 = part3 + (size_t)ks * n2 * nc;
    float* part1 = part2 + (size_t)ks * n1 * n2;
    if (ks > 1) {
        P.b3.ks = ks; P.b3.dW_part = part3;
        P.b2.ks = ks; P.b2.dW_part = part2;
        P.b1.ks = ks; P.b1.dW_part = part1;
    }
    // (folding Graph_BN's backward into the top layer's first backward kernel was measured: a wash at B = 256, +11 us at
    //  B = 1024 -- every packed row then gathers two molecule rows instead of one)
    P.bg = HeadGbn{B, F, sc.dgn, sv.g, sv.bn_g, P.sb_g, sc.dg, hg->d_gbn_w, hg->d_gbn_b, m->training};
    if (P.sync) { P.bg.cnt = cn_g; P.bg.gscale = gscale; }
    if (ks > 1) {
        P.bg.sum[0] = HeadDwSum{hg->d_den3_w, part3, n2 * nc, ks};
        P.bg.sum[1] = HeadDwSum{hg->d_den2_w, part2, n1 * n2, ks};
        P.bg.sum[2] = HeadDwSum{hg->d_den1_w, part1, F * n1, ks};
        P.bg.nsum = 3;
    }
    return P;
}

// pack -> layers -> read-out (everything in front of the head): fills sv / sc
static int model_forward_trunk(const eagcn_batch* b, const eagcn_model* m, const float* afm, const int64_t* size, void* saved,
                               size_t saved_bytes, void* scratch, size_t scratch_bytes, ModelSaved& sv, ModelScratch& sc,
                               void* stream, const char* who) {
    hipStream_t s = (hipStream_t)stream;
    RC(check_model(b, m, who));
    EAGCN_CHECK_GEMM3(who);
    EAGCN_CHECK_ARG((afm || m->input_packed) && saved && scratch, "%s: null buffer", who);
    EAGCN_CHECK_ARG(m->molfp_mode == 0 || size, "%s: 'ave' read-out needs size", who);
    EAGCN_CHECK_ARG(carve_saved(saved, b, m, &sv) <= saved_bytes, "%s: saved block too small", who);
    EAGCN_CHECK_ARG(carve_scratch(scratch, b, m, &sc) <= scratch_bytes, "%s: scratch too small", who);
    const eagcn_head_params* h = &m->head;
    // stream hand-offs (eagcn_model.start_signal / wait_flag): in the parameter-packing launch, which reads nothing of the batch --
    // unless the input is packed HERE, in front of it
    HandOff ho{m->start_signal, m->wait_flag, wait_err_word(), 200000000ull};
    if (!m->input_packed && (m->wait_flag || m->start_signal)) {
        handoff_kernel<<<1, 64, 0, s>>>(ho);
        EAGCN_LAUNCH_CHECK();
        ho.start = nullptr; ho.flag = nullptr;
    }
    ZeroJob zj;                                   // hand-off flags + the head's sums: cleared by the packing launch below
    RC(gemm3_zero_job(sc.layer, sc.layer_bytes, sc.hst, (HEAD_COPIES + 1) * sc.n_hst + HEAD_WS, &zj));
    if (!m->input_packed)
        RC(eagcn_pack_rows(b, afm, layout_width(&m->layer[0].in), &m->layer[0].in, sv.x0, stream));
    const float* x = sv.x0;
    {   // the parameters of every layer are re-laid by one launch
        const eagcn_layer_params* ps[4];
        void* pk[4];
        size_t pkb[4];
        for (int l = 0; l < m->n_layers; ++l) { ps[l] = &m->layer[l]; pk[l] = sv.L[l].packed; pkb[l] = sv.L[l].packed_bytes; }
        RC(pack_params_all(b, ps, pk, pkb, m->n_layers, stream, &zj, &ho));
    }
    for (int l = 0; l < m->n_layers; ++l) {
        LayerSaved& L = sv.L[l];
        eagcn_layer_bufs w;
        memset(&w, 0, sizeof(w));
        w.x = x; w.P = L.P; w.Y = L.Y; w.rscale = L.rscale; w.bn = L.bn; w.xout = L.xout; w.pad_row = L.pad_row;
        w.scratch = sc.layer; w.scratch_bytes = sc.layer_bytes; w.packed = L.packed; w.packed_bytes = L.packed_bytes;
        w.stats_hook = m->stats_hook; w.stats_user = m->stats_user;
        w.x_planes = l > 0 ? sv.L[l - 1].xout_planes : nullptr; w.xout_planes = L.xout_planes;
        // a hidden layer whose consumer reads plane images in both directions leaves no fp32 output (layer.hip bn_apply)
        const bool po = hidden_planes_only(b, m, sv, l);
        RC(layer_forward_impl(b, &m->layer[l], &w, stream, true, l == m->n_layers - 1 && fused_readout(m), po));
        x = po ? nullptr : L.xout;
    }
    const eagcn_layer_params* last = &m->layer[m->n_layers - 1];
    const LayerSaved& LL = sv.L[m->n_layers - 1];
    const eagcn_layout lay = out_layout(last);
    const int B = b->B, F = h->f_in, n1 = h->n_den1, n2 = h->n_den2;
    double *st_g = sc.hst, *st_1 = sc.hst + 2 * F + 2, *st_2 = sc.hst + 2 * (F + n1) + 4;
    const int copies = (m->stats_hook && m->training) ? 1 : HEAD_COPIES;
    if (fused_readout(m)) {
        // relu / dropout / mask of the top layer applied while the atoms are summed; Graph_BN's column sums in the same launch
        ReadoutBn rb;
        rb.Y = LL.Y; rb.ldy = LL.fp; rb.bn = LL.bn; rb.fp = LL.fp;
        fill_drop(last->dropout, last->training, &rb.do_drop, &rb.thr, &rb.inv_keep);
        rb.seed = last->seed; rb.seed_dev = last->seed_dev;
        rb.size = size; rb.mode = m->molfp_mode; rb.g = sv.g; rb.F = F; rb.st = st_g;
        rb.cnt0 = st_g + 2 * F; rb.cnt1 = st_1 + 2 * n1; rb.cnt2 = st_2 + 2 * n2;
        rb.st_copies = copies; rb.st_stride = sc.n_hst;
        RC(readout_bn_forward(b, &lay, rb, stream));
    } else if (pad_sampled(m))
        RC(readout_forward_sampled(b, LL.xout, &lay, last, LL.bn + (size_t)LL.fp /* shift row of the BatchNorm table */, size,
                                   m->molfp_mode, sv.g, F, sv.pad_cnt, sv.padc, sv.pad_tab, stream));
    else
        RC(eagcn_readout_forward(b, LL.xout, &lay, last->structure == EAGCN_STRUCT_WEIGHTED ? LL.pad_row : nullptr,
                                 size, m->molfp_mode, sv.g, F, stream));
    // (eagcn_model.fwd_signal: bumped by the head's first launch -- head_plan sets HeadFwd.signal; both callers launch it next)
    if (!fused_readout(m)) RC(head_colstats(sv.g, B, F, st_g, s, st_g + 2 * F, st_1 + 2 * n1, st_2 + 2 * n2));
    return EAGCN_OK;
}

static int stats_hook_call(const eagcn_model* m, bool sync, double* buf, int n, void* stream) {
    if (!sync) return EAGCN_OK;
    if (m->stats_hook(buf, n, stream, m->stats_user)) {
        set_error("eagcn_model: the sync-BatchNorm all-reduce hook failed");
        return EAGCN_ERR_HIP;
    }
    return EAGCN_OK;
}

// head forward as separate launches (head2.hip): every BatchNorm's sums come from the kernel that produces its input, its
// normalisation is applied by the product that consumes it
static int head_forward_launches(const eagcn_model* m, const HeadPlan& P, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    const eagcn_head_params* h = &m->head;
    RC(stats_hook_call(m, P.sync, P.st_g, 2 * h->f_in + 1, stream));
    RC(head_fwd(P.f1, s));
    RC(stats_hook_call(m, P.sync, P.st_1, 2 * h->n_den1 + 1, stream));
    RC(head_fwd(P.f2, s));
    RC(stats_hook_call(m, P.sync, P.st_2, 2 * h->n_den2 + 1, stream));
    RC(head_fwd(P.f3, s));
    return EAGCN_OK;
}
// ... and the head backward (sync-BatchNorm: the backward sums of every head BatchNorm are summed across the ranks before the
// launch that consumes them)
static int head_backward_launches(const eagcn_model* m, const HeadPlan& P, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    const eagcn_head_params* h = &m->head;
    RC(head_bwd(P.b3, s));
    RC(stats_hook_call(m, P.sync, P.sb_2, 2 * h->n_den2, stream));
    RC(head_bwd(P.b2, s));
    RC(stats_hook_call(m, P.sync, P.sb_1, 2 * h->n_den1, stream));
    RC(head_bwd(P.b1, s));
    RC(stats_hook_call(m, P.sync, P.sb_g, 2 * h->f_in, stream));
    RC(head_gbn_bwd(P.bg, s));
    return EAGCN_OK;
}
// read-out backward: evaluated inside the last layer's first backward kernel (ReadoutGrad); only the gradient of the common
// non-stored row of Weighted_sum needs a (tiny) launch of its own
static int readout_pad_backward(const eagcn_batch* b, const eagcn_model* m, const int64_t* size, const ModelSaved& sv,
                                const ModelScratch& sc, void* stream) {
    const eagcn_layer_params* last = &m->layer[m->n_layers - 1];
    const eagcn_layout lay = out_layout(last);
    const int F = m->head.f_in;
    if (pad_sampled(m)) return readout_backward_pad_views(b, sc.dg, &lay, size, m->molfp_mode, F, last->K, sv.pad_cnt, last->dropout, sc.dpad, stream);
    if (last->structure == EAGCN_STRUCT_WEIGHTED) return readout_backward_pad(b, sc.dg, &lay, size, m->molfp_mode, F, sc.dpad, stream);
    return EAGCN_OK;
}

extern "C" int eagcn_model_forward(const eagcn_batch* b, const eagcn_model* m, const float* afm,
                                   const int64_t* size, void* saved, size_t saved_bytes, void* scratch,
                                   size_t scratch_bytes, float* out, float* graph_rep, void* stream) {
    EAGCN_CHECK_ARG(out && graph_rep, "eagcn_model_forward: null buffer");
    ModelSaved sv;
    ModelScratch sc;
    RC(model_forward_trunk(b, m, afm, size, saved, saved_bytes, scratch, scratch_bytes, sv, sc, stream, "eagcn_model_forward"));
    const HeadPlan P = head_plan(b, m, sv, sc, out, graph_rep, nullptr, nullptr, nullptr);
    return head_forward_launches(m, P, stream);
}

// ---- the head alone (include/eagcn_hip.h eagcn_head_*): the GAT baseline and the Diff_Pooling read-out form the molecule
//      fingerprints through the layer-level entry points and hand them over here; stages and launch order are head_plan's ----------
namespace eagcn {
static int check_head(const eagcn_head_params* h, int B, int training, const char* who) {
    EAGCN_CHECK_ARG(h, "%s: null argument", who);
    EAGCN_CHECK_ARG(B >= 1 && h->f_in >= 1 && h->n_den1 >= 1 && h->n_den2 >= 1 && h->nclass >= 1, "%s: bad head sizes", who);
    EAGCN_CHECK_ARG(h->den1_w && h->den2_w && h->den3_w && h->gbn_w && h->gbn_b && h->gbn_rm && h->gbn_rv && h->bn1_w && h->bn1_b &&
                        h->bn1_rm && h->bn1_rv && h->bn2_w && h->bn2_b && h->bn2_rm && h->bn2_rv, "%s: null head parameter", who);
    EAGCN_CHECK_ARG(!training || B > 1, "%s: BatchNorm in training mode needs more than one molecule", who);
    return EAGCN_OK;
}
static size_t carve_head_saved(void* base, const eagcn_head_params* h, int B, ModelSaved* out) {
    Carver2 c(base);
    ModelSaved s;
    memset(&s, 0, sizeof(s));
    s.h1 = c.take<float>((size_t)B * h->n_den1);
    s.h2 = c.take<float>((size_t)B * h->n_den2);
    s.bn_g = c.take<float>((size_t)4 * h->f_in);
    s.bn_1 = c.take<float>((size_t)4 * h->n_den1);
    s.bn_2 = c.take<float>((size_t)4 * h->n_den2);
    if (out) *out = s;
    return c.off;
}
static size_t carve_head_scratch(void* base, const eagcn_head_params* h, int B, ModelScratch* out) {
    Carver2 c(base);
    ModelScratch s;
    memset(&s, 0, sizeof(s));
    s.n_hst = 2 * (h->f_in + h->n_den1 + h->n_den2) + 6;
    s.hst = c.take<double>((size_t)(HEAD_COPIES + 1) * s.n_hst + HEAD_WS);
    s.hsb = s.hst + (size_t)HEAD_COPIES * s.n_hst;
    s.hws = s.hsb + s.n_hst;
    s.da2 = c.take<float>((size_t)B * h->n_den2);
    s.da1 = c.take<float>((size_t)B * h->n_den1);
    s.dgn = c.take<float>((size_t)B * h->f_in);
    s.hdw_ks = head_dw_chunks(B);
    s.hdw = c.take<float>(s.hdw_ks > 1 ? (size_t)s.hdw_ks * ((size_t)h->f_in * h->n_den1 + (size_t)h->n_den1 * h->n_den2 +
                                                             (size_t)h->n_den2 * h->nclass) : 1);
    if (out) *out = s;
    return c.off;
}
// (a kernel, not hipMemsetAsync: inside a captured step every node of the chain is then a kernel node)
__global__ __launch_bounds__(256) void head_zero_kernel(double* __restrict__ p, int n) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) p[i] = 0.0;
}
static int head_zero(double* p, size_t n, hipStream_t s) {
    head_zero_kernel<<<(unsigned)std::max<size_t>(1, std::min<size_t>((n + 255) / 256, 64)), 256, 0, s>>>(p, (int)n);
    EAGCN_LAUNCH_CHECK();
    return EAGCN_OK;
}
// the pieces of eagcn_batch / eagcn_model that head_plan reads
static void head_alone_model(const eagcn_head_params* h, int B, int training, uint64_t seed, const uint64_t* seed_dev,
                             eagcn_batch* b, eagcn_model* m) {
    memset(b, 0, sizeof(*b));
    memset(m, 0, sizeof(*m));
    b->B = B;
    m->head = *h; m->training = training ? 1 : 0; m->head_seed = seed; m->head_seed_dev = seed_dev;
}
}  // namespace eagcn

extern "C" size_t eagcn_head_saved_bytes(const eagcn_head_params* h, int B) { return h ? carve_head_saved(nullptr, h, B, nullptr) : 0; }
extern "C" size_t eagcn_head_scratch_bytes(const eagcn_head_params* h, int B) { return h ? carve_head_scratch(nullptr, h, B, nullptr) : 0; }

extern "C" int eagcn_head_forward(const eagcn_head_params* h, int B, int training, uint64_t seed, const uint64_t* seed_dev,
                                  const float* g, void* saved, size_t saved_bytes, void* scratch, size_t scratch_bytes, float* out,
                                  float* graph_rep, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    RC(check_head(h, B, training, "eagcn_head_forward"));
    EAGCN_CHECK_GEMM3("eagcn_head_forward");
    EAGCN_CHECK_ARG(g && saved && scratch && out && graph_rep, "eagcn_head_forward: null buffer");
    ModelSaved sv;
    ModelScratch sc;
    EAGCN_CHECK_ARG(carve_head_saved(saved, h, B, &sv) <= saved_bytes, "eagcn_head_forward: saved block too small");
    EAGCN_CHECK_ARG(carve_head_scratch(scratch, h, B, &sc) <= scratch_bytes, "eagcn_head_forward: scratch too small");
    eagcn_batch b;
    eagcn_model m;
    head_alone_model(h, B, training, seed, seed_dev, &b, &m);
    sv.g = const_cast<float*>(g);                   // (read only: the stages take the fingerprints where the caller holds them)
    const int F = h->f_in, n1 = h->n_den1, n2 = h->n_den2;
    RC(head_zero(sc.hst, (size_t)(HEAD_COPIES + 1) * sc.n_hst + HEAD_WS, s));
    double *st_g = sc.hst, *st_1 = sc.hst + 2 * F + 2, *st_2 = sc.hst + 2 * (F + n1) + 4;
    RC(head_colstats(g, B, F, st_g, s, st_g + 2 * F, st_1 + 2 * n1, st_2 + 2 * n2));
    const HeadPlan P = head_plan(&b, &m, sv, sc, out, graph_rep, nullptr, nullptr, nullptr);
    return head_forward_launches(&m, P, stream);
}

extern "C" int eagcn_head_backward(const eagcn_head_params* h, int B, int training, uint64_t seed, const uint64_t* seed_dev,
                                   const float* g, void* saved, size_t saved_bytes, void* scratch, size_t scratch_bytes,
                                   const float* dout, const float* dgraph_rep, const eagcn_head_grads* hg, float* dg, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    RC(check_head(h, B, training, "eagcn_head_backward"));
    EAGCN_CHECK_GEMM3("eagcn_head_backward");
    EAGCN_CHECK_ARG(g && saved && scratch && dout && hg && dg, "eagcn_head_backward: null buffer");
    EAGCN_CHECK_ARG(hg->d_den1_w && hg->d_den2_w && hg->d_den3_w && hg->d_gbn_w && hg->d_gbn_b && hg->d_bn1_w && hg->d_bn1_b &&
                        hg->d_bn2_w && hg->d_bn2_b, "eagcn_head_backward: null head gradient");
    ModelSaved sv;
    ModelScratch sc;
    EAGCN_CHECK_ARG(carve_head_saved(saved, h, B, &sv) <= saved_bytes, "eagcn_head_backward: saved block too small");
    EAGCN_CHECK_ARG(carve_head_scratch(scratch, h, B, &sc) <= scratch_bytes, "eagcn_head_backward: scratch too small");
    eagcn_batch b;
    eagcn_model m;
    head_alone_model(h, B, training, seed, seed_dev, &b, &m);
    sv.g = const_cast<float*>(g);
    sc.dg = dg;
    // (the scratch is transient: nothing of the forward call is expected in it; the backward sums start from zero)
    RC(head_zero(sc.hsb, (size_t)sc.n_hst, s));
    const HeadPlan P = head_plan(&b, &m, sv, sc, nullptr, nullptr, dout, dgraph_rep, hg);
    return head_backward_launches(&m, P, stream);
}

namespace eagcn {
__global__ void scale_loss_kernel(float* __restrict__ loss, float* __restrict__ dout, int n, const float* __restrict__ scale) {
    const float sc = *scale;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dout[i] *= sc;
    if (i == 0) loss[0] *= sc;
}
}  // namespace eagcn

// Forward, loss and the HEAD's backward of a training step: after this call d(loss)/d(molecule fingerprints) and every head
// gradient exist; the caller continues with eagcn_model_backward_range(with_head = 0, ...).  The head runs as ONE launch
// (head2.hip head_all_kernel) wherever its conditions hold, as the separate launches otherwise -- same results.
extern "C" int eagcn_model_forward_step(const eagcn_batch* b, const eagcn_model* m, const float* afm, const int64_t* size,
                                        void* saved, size_t saved_bytes, void* scratch, size_t scratch_bytes, float* out,
                                        float* graph_rep, const eagcn_step_loss* loss, const float* dgraph_rep,
                                        const eagcn_head_grads* hg, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    EAGCN_CHECK_ARG(out && graph_rep && loss && hg, "eagcn_model_forward_step: null argument");
    EAGCN_CHECK_ARG(m && m->training, "eagcn_model_forward_step: a training-mode model is required");
    EAGCN_CHECK_ARG((loss->kind == 0 || loss->kind == 1) && loss->labels && loss->loss && loss->dout && (loss->kind == 1 || loss->class_weight),
                    "eagcn_model_forward_step: bad loss descriptor");
    EAGCN_CHECK_ARG(hg->d_den1_w && hg->d_den2_w && hg->d_den3_w && hg->d_gbn_w && hg->d_gbn_b && hg->d_bn1_w &&
                        hg->d_bn1_b && hg->d_bn2_w && hg->d_bn2_b, "eagcn_model_forward_step: null head gradient");
    ModelSaved sv;
    ModelScratch sc;
    RC(model_forward_trunk(b, m, afm, size, saved, saved_bytes, scratch, scratch_bytes, sv, sc, stream, "eagcn_model_forward_step"));
    const HeadPlan P = head_plan(b, m, sv, sc, out, graph_rep, loss->dout, dgraph_rep, hg);
    const eagcn_head_params* h = &m->head;
    if (!P.sync && head_mid_ok(h->n_den2, h->nclass)) {
        // F1, F2 (counts the labelled entries on the way), then out + loss + d a2 as ONE launch, dense 3's weight gradient in
        // dense 2's backward launch: six launches instead of eight
        HeadPlan Q = P;
        unsigned* words = reinterpret_cast<unsigned*>(sc.hws);
        if (loss->kind == 0) { Q.f2.lab = loss->labels; Q.f2.nlab = b->B * h->nclass; Q.f2.lab_cnt = words + 1; }
        RC(head_fwd(Q.f1, s));
        RC(head_fwd(Q.f2, s));
        HeadMid M;
        M.f3 = Q.f3; M.b3 = Q.b3;
        M.L = HeadLoss{loss->kind, loss->labels, loss->class_weight, loss->loss, loss->scale, loss->dout};
        M.ws = sc.hws;
        RC(head_mid(M, s));
        RC(head_bwd_pair(Q.b2, Q.b3, HeadLossFin{loss->loss, sc.hws, loss->scale, loss->kind, b->B * h->nclass}, s));
        RC(head_bwd(Q.b1, s));
        RC(head_gbn_bwd(Q.bg, s));
    } else {
        RC(head_forward_launches(m, P, stream));
        if (loss->kind == 0) RC(eagcn_bce_loss(out, loss->labels, loss->class_weight, b->B, h->nclass, loss->loss, loss->dout, stream));
        else RC(eagcn_mse_loss(out, loss->labels, b->B * h->nclass, loss->loss, loss->dout, stream));
        if (loss->scale) {
            const int n = b->B * h->nclass;
            scale_loss_kernel<<<cdiv(n, 256), 256, 0, s>>>(loss->loss, loss->dout, n, loss->scale);
            EAGCN_LAUNCH_CHECK();
        }
        RC(head_backward_launches(m, P, stream));
    }
    return readout_pad_backward(b, m, size, sv, sc, stream);
}

extern "C" int eagcn_model_backward(const eagcn_batch* b, const eagcn_model* m, const int64_t* size, void* saved,
                                    size_t saved_bytes, void* scratch, size_t scratch_bytes, const float* dout,
                                    const float* dgraph_rep, const eagcn_layer_grads* lg,
                                    const eagcn_head_grads* hg, void* stream) {
    EAGCN_CHECK_ARG(m, "eagcn_model_backward: null model");
    return eagcn_model_backward_range(b, m, size, saved, saved_bytes, scratch, scratch_bytes, dout, dgraph_rep, lg, hg, 1,
                                      m->n_layers - 1, 0, stream);
}

extern "C" int eagcn_model_backward_range(const eagcn_batch* b, const eagcn_model* m, const int64_t* size, void* saved,
                                          size_t saved_bytes, void* scratch, size_t scratch_bytes, const float* dout,
                                          const float* dgraph_rep, const eagcn_layer_grads* lg,
                                          const eagcn_head_grads* hg, int with_head, int layer_hi, int layer_lo, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    RC(check_model(b, m, "eagcn_model_backward"));
    EAGCN_CHECK_GEMM3("eagcn_model_backward");
    EAGCN_CHECK_ARG(saved && scratch && dout && lg && hg, "eagcn_model_backward: null buffer");
    EAGCN_CHECK_ARG(hg->d_den1_w && hg->d_den2_w && hg->d_den3_w && hg->d_gbn_w && hg->d_gbn_b && hg->d_bn1_w &&
                        hg->d_bn1_b && hg->d_bn2_w && hg->d_bn2_b, "eagcn_model_backward: null head gradient");
    EAGCN_CHECK_ARG(layer_hi < m->n_layers && layer_lo >= 0 && layer_lo <= layer_hi + 1,
                    "eagcn_model_backward_range: layers %d..%d of %d", layer_hi, layer_lo, m->n_layers);
    EAGCN_CHECK_ARG(!with_head || layer_hi == m->n_layers - 1 || layer_hi < layer_lo,
                    "eagcn_model_backward_range: the head's backward is followed by the top layer");
    ModelSaved sv;
    ModelScratch sc;
    EAGCN_CHECK_ARG(carve_saved(saved, b, m, &sv) <= saved_bytes, "eagcn_model_backward: saved block too small");
    EAGCN_CHECK_ARG(carve_scratch(scratch, b, m, &sc) <= scratch_bytes, "eagcn_model_backward: scratch too small");
    const int F = m->head.f_in;
    // (no clearing launch: the forward call left the hand-off flags and the backward sums zero; owners reset their flags)
    const ZeroJob zb{nullptr, 0, sc.hsb, sc.n_hst};
    hipStream_t side = m->aux_stream ? (hipStream_t)m->aux_stream : s;
    const bool forked = side != s;
    const eagcn_layer_params* last = &m->layer[m->n_layers - 1];
    const eagcn_layout lay = out_layout(last);
    const bool weighted = last->structure == EAGCN_STRUCT_WEIGHTED;
    const bool sampled = pad_sampled(m);
    if (with_head) {
        const HeadPlan P = head_plan(b, m, sv, sc, nullptr, nullptr, dout, dgraph_rep, hg);
        RC(head_backward_launches(m, P, stream));
        RC(readout_pad_backward(b, m, size, sv, sc, stream));
    }
    ReadoutGrad rgd;
    memset(&rgd, 0, sizeof(rgd));
    rgd.dg = sc.dg; rgd.F = F; rgd.size = size; rgd.mode = m->molfp_mode; rgd.map = make_colmap(&lay);
    const int top_l = m->n_layers - 1;
    EdgeDrain pend_in, pend_out;
    pend_in.eacc = nullptr;
    for (int l = layer_hi; l >= layer_lo; --l) {
        // d(layer input) ping-pongs between two buffers, top-down: the layer t = top - l below the top reads the buffer the
        // layer above wrote (t odd: dxb, t even: dxa) and writes the other one
        const int t = top_l - l;
        float* cur = (t & 1) ? sc.dxb : sc.dxa;
        float* other = (t & 1) ? sc.dxa : sc.dxb;
        LayerSaved& L = sv.L[l];
        eagcn_layer_bufs w;
        memset(&w, 0, sizeof(w));
        w.x = l == 0 ? sv.x0 : (hidden_planes_only(b, m, sv, l - 1) ? nullptr : sv.L[l - 1].xout);
        w.P = L.P; w.Y = L.Y; w.rscale = L.rscale; w.bn = L.bn; w.xout = L.xout; w.pad_row = L.pad_row;
        w.scratch = sc.layer; w.scratch_bytes = sc.layer_bytes; w.packed = L.packed; w.packed_bytes = L.packed_bytes;
        w.aux_stream = m->aux_stream;
        w.stats_hook = m->stats_hook; w.stats_user = m->stats_user;
        w.x_planes = l > 0 ? sv.L[l - 1].xout_planes : nullptr;
        const bool top = l == top_l;
        const float* dpad = (weighted && top) ? sc.dpad : nullptr;
        // (a layer that has only its edge-gradient reduction left hands it to the next layer's first kernel -- if one follows in
        //  this call)
        RC(layer_backward_impl(b, &m->layer[l], &w, top ? nullptr : cur, top ? &rgd : nullptr, dpad,
                               l > 0 ? other : nullptr, &lg[l], stream, top && sampled, top ? &zb : nullptr,
                               pend_in.eacc ? &pend_in : nullptr, l > layer_lo ? &pend_out : nullptr));
        pend_in = pend_out;
        if (l == layer_lo) pend_in.eacc = nullptr;
    }
    if (forked) RC(stream_after(s, side));                       // join: every gradient is complete on s
    return EAGCN_OK;
}

extern "C" int eagcn_stream_signal_flag(uint32_t* flag, void* stream) {
    EAGCN_CHECK_ARG(flag != nullptr, "eagcn_stream_signal_flag: null flag");
    eagcn::signal_flag_kernel<<<1, 64, 0, (hipStream_t)stream>>>(flag);
    EAGCN_LAUNCH_CHECK();
    return EAGCN_OK;
}
extern "C" int eagcn_stream_wait_flag(uint32_t* flag, double budget_seconds, void* stream) {
    EAGCN_CHECK_ARG(flag != nullptr && budget_seconds > 0.0, "eagcn_stream_wait_flag: null flag / no budget");
    return eagcn::launch_wait_flag(flag, budget_seconds, (hipStream_t)stream);
}
extern "C" int eagcn_stream_wait_timeouts(void) {
    return eagcn::g_wait_err_host ? __atomic_load_n(eagcn::g_wait_err_host, __ATOMIC_RELAXED) : 0;
}
extern "C" void eagcn_stream_wait_reset(void) {
    if (eagcn::g_wait_err_host) __atomic_store_n(eagcn::g_wait_err_host, 0, __ATOMIC_RELAXED);
}

extern "C" int eagcn_stream_wait_counter(const uint32_t* counter, uint32_t value, void* stream) {
    EAGCN_CHECK_ARG(counter != nullptr, "eagcn_stream_wait_counter: null counter");
    eagcn::wait_counter_kernel<<<1, 64, 0, (hipStream_t)stream>>>(counter, value, eagcn::wait_err_word(), 200000000ull);   // 2 s
    EAGCN_LAUNCH_CHECK();
    return EAGCN_OK;
}
