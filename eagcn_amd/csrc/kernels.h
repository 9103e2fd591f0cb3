// Argument blocks and launchers shared between the HIP translation units.
#pragma once
#include "bx3.h"
#include "common.h"

namespace eagcn {

struct AggArgs {
    eagcn_batch bt;
    ViewCols vc;
    const float* src; int lds;   // P (forward) or dY' (transposed) [T][lds]
    float* dst; int ldd;         // Y' (forward) or dP (transposed)
    const float* sig;            // [K][256] sigmoid(att weight) by bond code, entry 0 = 0
    const float* rsig;           // [K] sigmoid(self_r)
    float* rscale;               // [K][T] m_i/rowsum_i: written by forward, read by transposed
    double* stats;               // forward: [grid.x][Fp][2] partial (sum y, sum y^2)
    int nchunk;
    int cpw = 1;                 // lagg.hip: consecutive 32-column chunks a workgroup takes per block (nchunk then counts chunk GROUPS)
    int xcd = 0;                 // wave-per-tile variant: tiles handed out so that an XCD works on CONTIGUOUS tiles (agg.hip)
    BxOut planes = {nullptr, 0, 0, 0};   // transposed: dP leaves as bf16 planes INSTEAD of the fp32 matrix (only the plane GEMMs read it)
    // transposed, lagg.hip only: `src` holds dH (the BatchNorm's upstream gradient), not dY': the BatchNorm backward's second pass
    //     dY' = sc (dH - c1 - (Y' - mu) inv c2)      (layer.hip bn_bwd_apply_kernel)
    // is evaluated while the rows are staged (bn = the layer's [4][Fp] table, cc = [2][Fp] {c1, c2}, Y' = EdgeArgs.Y); null: src is dY'
    const float* bn_tab = nullptr; const float* bn_cc = nullptr; int bn_fp = 0;
    // ... and for a Weighted_sum layer (w_aw != null, round 6) `src` is the layer's UPSTREAM gradient [T][lds], K times narrower than dH:
    //     dH[r][off_k + f] = src[r][f] ave_w[off_k + f] dropout(r, off_k + f) [relu'(sc Y' + sh) > 0]   (layer.hip bn_bwd_reduce_kernel<true,...>)
    // is re-formed in the staging as well -- neither dH nor dY' of the layer (T x K F floats each) ever exists in memory
    const float* w_aw = nullptr;                 // [Fp] the view weight per packed column (colp row CP_AVEW)
    int w_drop = 0; uint32_t w_thr = 0; float w_inv_keep = 1.0f; uint64_t w_seed = 0; const uint64_t* w_seed_dev = nullptr;
};
int agg_grid_x(const eagcn_batch* b);
bool agg_ksplit(const eagcn_batch* b);
int launch_agg(AggArgs a, bool trans, hipStream_t s);
// the same operator over the bond lists of the index, LDS-staged (lagg.hip): a workgroup owns a ROW BLOCK of whole molecules (eagcn_batch.blk) x a 32-column chunk of one view
constexpr int LAGG_RB = 256;                 // packed rows per block (= the largest molecule the path takes)
constexpr int LAGG_MAXM = 16;                // molecules per block
struct EdgeArgs;
int lagg_parts();
bool lagg_wfuse();                           // Weighted_sum layers: the transposed kernel re-forms dH from the upstream gradient (EAGCN_LAGG_WFUSE=0: off)
bool lagg_use(const eagcn_batch* b, int dir, bool absorbs_bn, const ViewCols* vc = nullptr);   // this batch takes that path (policy per direction + the index carries bond lists and row blocks)
bool lagg_wanted(int B, int N, int structure);
int lagg_block_rows(const eagcn_batch* b);     // rows per row block for batches of this shape (index_blocks_kernel; <= LAGG_RB)
int lagg_slabs(const eagcn_batch* b);        // capacity of its BatchNorm partial slabs (one per row block; meta[NBLK] of them are live)
int launch_lagg_fwd(AggArgs a, hipStream_t s);
int launch_lagg_bwd(AggArgs a, const EdgeArgs& e, hipStream_t s);   // transposed aggregation + edge gradients (e.atomic must be set)

struct EdgeArgs {
    eagcn_batch bt;
    ViewCols vc;
    const float* dY; const float* Y; const float* P; int ld;
    const float* sig; const float* rsig; const float* rscale;
    double* datt;                // [grid.x][K][EDGE_SLAB] per-workgroup partials: 256 codes + self term ...
    int atomic;                  // ... or (atomic != 0) [EDGE_COPIES][K][EDGE_SLAB] shared accumulators (fp64 atomics)
    int xcd = 0;                 // row blocks handed out so that an XCD works on CONTIGUOUS packed rows (agg.hip edge_grad_body)
};
constexpr int EDGE_SLAB = 264;
// The edge-gradient workgroups add their bond-type histograms (a dozen non-zero bins each) to one of EDGE_COPIES fp64 accumulator
// slabs instead of writing a slab of their own: the reduction that follows reads 8 slabs, not one per workgroup (12.7 MB at
// batch 1024).  The accumulators live right behind the GEMM hand-off workspace in every layer scratch carving -- same place for
// every layer and direction, never used for anything else --, are cleared together with the hand-off flags once per API call
// and zeroed again by the reduction that drains them.
constexpr int EDGE_COPIES = 8;
inline size_t edge_acc_bytes() { return align256((size_t)EDGE_COPIES * EAGCN_MAX_VIEWS * EDGE_SLAB * sizeof(double)); }
int edge_grid_x(const eagcn_batch* b);
int launch_edge_grad(const EdgeArgs& a, hipStream_t s);
int launch_agg_edge(AggArgs a, const EdgeArgs& e, hipStream_t s);   // transposed aggregation + edge gradients, one grid

enum { BN_SC = 0, BN_SH, BN_MU, BN_INV };     // rows of a layer's [4][Fp] BatchNorm coefficient table

struct ColMapD {                 // exact <-> packed column map of a layout
    int nseg;
    int w[EAGCN_MAX_SEGS], p[EAGCN_MAX_SEGS];
};
inline ColMapD make_colmap(const eagcn_layout* l) {
    ColMapD m;
    m.nseg = l->nseg;
    for (int i = 0; i < EAGCN_MAX_SEGS; ++i) {
        m.w[i] = i < l->nseg ? l->width[i] : 0;
        m.p[i] = i < l->nseg ? l->pad[i] : 0;
    }
    return m;
}

// Upstream gradient of the LAST layer given per molecule instead of per row: the read-out backward
// (dx[r] = dg[mol(r)] / size, reference models.py:108-111) evaluated inside the layer's first backward
// kernel, so the [T][ldo] broadcast is never written or read.
struct ReadoutGrad {
    const float* dg; int F;      // [B][F] gradient of the molecule fingerprints
    const int64_t* size; int mode;
    ColMapD map;                 // layout of the layer output (packed column -> exact column)
};
int launch_readout_bwd_rows(const eagcn_batch* b, const ReadoutGrad& rg, int ld, float* dx, hipStream_t s);   // readout.hip
// ---- wave-autonomous balanced GEMM (gemm3.hip): NT (ta=0,tb=1) and TN (ta=1,tb=0) forms --------------------------------
struct G2Prob {                  // one product C[M,N] = op(A).op(B) in one of the three operand forms of a layer
    const float* A; const float* B; float* C;
    int lda, ldb, ldc;
    int M, N, K;                 // static extents (capacities where a device-side count exists)
    const int* M_dev;            // actual M (rows of A and C), read on the device
    const int* K_dev;            // actual K
};
struct DwScatter {               // epilogue of a layer's dW product: packed (input column, output column) -> blockK.graph_conv.weight.grad
    float* dW[EAGCN_MAX_VIEWS];
    ViewCols vc;
    ColMapD in;
    int in_identity;             // the input layout has no padded columns: packed row == exact row (set by the launchers)
};
// ---- XCD-local schedule of a paired launch (dX = dP.W^T with dW = X^T.dP) -------------------------------------------------
// The linear iteration space of the pair is laid out as eight SEGMENTS, one per XCD (dispatch slot b runs on XCD b % 8 and
// the kernel hands every XCD a contiguous run of wave ranges): segment x = the dX tiles of the row blocks [a_x, a_x+1) followed
// by, for EVERY dW tile, the k-steps [c_x, c_x+1) -- i.e. the long reduction of dW (K = packed rows) is split ACROSS the XCDs
// and each XCD only ever touches the rows [~x T/8, ~(x+1) T/8) of dP and X (both products!): they cross the fabric once
// and stay in that XCD's 4 MB L2, instead of every XCD streaming full-K column strips for "its" dW tiles.  The price is one
// partial dW slab per XCD (plain tile stores), summed by the gradient-unpacking launch that follows anyway.  The boundaries
// are a pure function of the device-side extents, shared by the kernel and that reduction.
constexpr int G3_XSEG = 8;
struct G3Plan {
    int a[G3_XSEG + 1];          // dX row-block boundaries
    int c[G3_XSEG + 1];          // dW k-step boundaries
};
__host__ __device__ inline void g3_plan(int RB, int CT0, int ipt0, int T1, int ipt1, int G, G3Plan& pl) {
    const long total = (long)RB * CT0 * ipt0 + (long)T1 * ipt1;
    const int R = 4 * G;
    const long per = total / R, rem = total - per * R;
    const int qn = G >> 3, rn = G & 7;
    pl.a[0] = 0;
    pl.c[0] = 0;
    for (int x = 1; x < G3_XSEG; ++x) {
        const long L = 4L * (x * qn + (x < rn ? x : rn));            // first wave range of XCD x
        const long tgt = L * per + (L < rem ? L : rem);              // iteration where that range starts
        int a = (int)(((long)RB * L + R / 2) / R);                   // row blocks in proportion to the waves in front
        a = a < pl.a[x - 1] ? pl.a[x - 1] : (a > RB ? RB : a);
        long cc = T1 > 0 ? (tgt - (long)a * CT0 * ipt0 + T1 / 2) / T1 : 0;
        if (tgt - (long)a * CT0 * ipt0 < 0) cc = 0;
        int c = (int)(cc > ipt1 ? ipt1 : cc);
        c = c < pl.c[x - 1] ? pl.c[x - 1] : c;
        pl.a[x] = a;
        pl.c[x] = c;
    }
    pl.a[G3_XSEG] = RB;
    pl.c[G3_XSEG] = ipt1;
}
size_t gemm3_workspace_bytes();
bool gemm3_xk_enabled();          // the schedule above is used for the paired backward products (EAGCN_GEMM3_XK=0 turns it off)
int gemm3_grid();
// the hand-off flags must be zero when a launch starts; every launch leaves them zero again, so ONE clear at the start of
// an API call (a tiny kernel, also under capture) covers all its launches on the same workspace
// (`zero`: optionally `nzero` doubles cleared by the same launch -- the head's BatchNorm sums)
int gemm3_clear_flags(void* workspace, size_t bytes, hipStream_t s, double* zero = nullptr, int nzero = 0);
int gemm3_failed();              // sticky, host-visible: a hand-off of some earlier launch timed out (its tile is NaN)
#define EAGCN_CHECK_GEMM3(who)                                                                                         \
    do {                                                                                                               \
        if (::eagcn::gemm3_failed()) {                                                                                 \
            ::eagcn::set_error("%s: a stream-K hand-off of an earlier GEMM launch timed out (a contributor wave was not " \
                               "co-resident with its owner): that launch's results are NaN-poisoned; call "           \
                               "eagcn_gemm_sk_reset_failed() after fixing the cause (EAGCN_GEMM3_WGS, concurrent work)", who); \
            return EAGCN_ERR_HIP;                                                                                      \
        }                                                                                                              \
    } while (0)
// words another kernel clears on the way (instead of a launch of its own): the hand-off flags of gemm3.hip and fp64 sums
int gemm_mode();             // 0 fp32 MFMA | 1 exact bf16 x 6, split by the consumer | 2 plain bf16 operands (gemm.hip)
                             // 3 exact bf16 x 3 PLANES written by the producers | 4 ONE bf16 plane (gemm_bx3.hip)
inline int gemm_planes() { const int m = gemm_mode(); return m == 3 ? 3 : m == 4 ? 1 : 0; }
struct ZeroJob { unsigned* u; int nu; double* d; int nd; };
// stream hand-offs that ride in a step's first launch (eagcn_model.start_signal / wait_flag): +1 on `start`, then a poll of `flag`
// until it is non-zero (cleared afterwards); after `budget` ticks of the 100 MHz clock the poll gives up and raises *err
struct HandOff { uint32_t* start; uint32_t* flag; int* err; unsigned long long budget; };
__device__ __forceinline__ void handoff_body(const HandOff& h) {
    if (h.start) __hip_atomic_fetch_add(h.start, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    if (h.flag) {
        const unsigned long long t0 = wall_clock64();
        while (__hip_atomic_load(h.flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) == 0u) {
            __builtin_amdgcn_s_sleep(8);
            if (wall_clock64() - t0 > h.budget) {
                if (h.err) __hip_atomic_store(h.err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                break;
            }
        }
        __hip_atomic_store(h.flag, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}
int gemm3_zero_job(void* workspace, size_t bytes, double* zero, int nzero, ZeroJob* out);
// zero-fill by a kernel (memset NODES of a captured graph are not ordered with their neighbours on replay, ROCm 7.x): for every
// clear that can sit inside a captured sequence
int zero_fill(void* p, size_t bytes, hipStream_t s);
bool gemm3_ok(const GemmDesc& g);
int launch_gemm3(const GemmDesc& g, const DwScatter* sc, void* workspace, size_t bytes, hipStream_t s);
// xk_slab > 0: XCD-local schedule, dW leaves as G3_XSEG partial slabs dw.C + x * xk_slab (sc must be null)
int launch_gemm3_pair(const GemmDesc& dx, const GemmDesc& dw, const DwScatter* sc, void* workspace, size_t bytes, hipStream_t s,
                      size_t xk_slab = 0);

// ---- model head (head2.hip) -----------------------------------------------------------------------------------------------
constexpr int HEAD_COPIES = 8;  // replicas of the head's forward BatchNorm sums (one with sync-BatchNorm: the hook reduces ONE block)
struct HeadDrop { int on; uint32_t thr; float inv_keep; uint64_t seed; const uint64_t* seed_dev; };
struct HeadFwd {                 // y = act(bn(x)) . W  (+ column sums of y)
    int B, K, N;
    const float* x; const double* st_in; const float *gamma, *beta; float *run_mean, *run_var; float* bn;
    const float* W; float* y; float* y2; double* st_out;
    int training, relu; float eps, momentum; HeadDrop drop;
    const double* cnt_in = nullptr;      // sync-BatchNorm: rows of the BatchNorm over ALL ranks (st_in holds their summed sums)
    // the sums live in st_copies replicas, st_stride doubles apart: a producer workgroup adds to replica (row tile % copies),
    // the consumer adds the replicas up (fp64 atomics on one 128-byte line are served one after the other, ~11 ns each: 1024
    // arrivals per line at B = 1024 kept den1 / den2 waiting 11 us each)
    int st_copies = 1, st_stride = 0;
    // optional job on the way: count the entries of lab[0 .. nlab) that are 0 or 1 (the batch's labelled entries of the BCE loss)
    // into *lab_cnt (zero on entry), every workgroup its share
    const float* lab = nullptr; int nlab = 0; unsigned* lab_cnt = nullptr;
    int nw = 0;                          // waves per workgroup that split the reduction: 4 / 8, 0 = by the launch's tile count
    uint32_t* signal = nullptr;          // optional: a device word bumped by the launch's first workgroup (eagcn_model.fwd_signal: "the forward
                                         // has passed its read-out" -- a launch of its own cost the step 4.6 us on the main stream)
};
struct HeadBwd {                 // backward of  y = act(bn_p(x)) . W : d(bn_p output) with its sums, and dW
    int B, K, N;
    const float* x; const float* bnp; int relu_p; HeadDrop drop; const float* W;
    const float* dy; const float* y; const float* bny; const double* sb_y; const float* extra;
    float *dgamma_y, *dbeta_y; float* dyp; double* sb_p; float* dW; int training;
    const double* cnt_y = nullptr;       // sync-BatchNorm: global row count of the BatchNorm behind y (sb_y: sums of all ranks)
    float gscale = 1.0f;                 // ... and 1 / world: d gamma / d beta = this rank's share of the summed sums
    // ks > 1: the weight gradient's reduction over the B batch rows is cut into ks row chunks, one workgroup per (tile, chunk);
    // chunk c's partial goes to dW_part + c * K * N and head_gbn_bwd adds the partials up in chunk order (a tile's whole
    // reduction in one workgroup is a chain of B / 128 dependent load batches: 8 at B = 1024, the longest pole of the launch)
    int ks = 1; float* dW_part = nullptr;
    int nw = 0;                          // (as HeadFwd.nw)
};
inline int head_dw_chunks(int B) { return B > 256 ? (B + 255) / 256 < 16 ? (B + 255) / 256 : 16 : 1; }
struct HeadDwSum { float* dst; const float* part; int n, ks; };     // dst[i] = sum_c part[c * n + i]
struct HeadGbn { int B, F; const float *dgn, *g, *bn; const double* sb; float *dg, *dgamma, *dbeta; int training;
                 const double* cnt = nullptr; float gscale = 1.0f;
                 HeadDwSum sum[3] = {}; int nsum = 0; };            // partial weight gradients of the dense layers (HeadBwd.ks)
// the middle of a TRAINING step's head in one launch: last forward stage, loss, dense 3's d(input) (head2.hip head_mid_kernel)
struct HeadLoss { int kind;                 // 0: weighted BCE with logits over the labelled entries (train.py:326-331), 1: MSE (train.py:321-325)
                  const float* labels; const float* weight; float* loss; const float* scale; float* dout; };
struct HeadMid { HeadFwd f3; HeadBwd b3; HeadLoss L;
                 double* ws; };            // HEAD_WS doubles, zero on entry: {unused, labelled entries} as two 32-bit words, loss sum
constexpr int HEAD_WS = 4;
struct HeadLossFin { float* loss; const double* ws; const float* scale; int kind, n; };   // loss value = ws[1] / count (* scale); loss == null: no job
bool head_mid_ok(int n_den2, int nclass);
int head_mid(const HeadMid& a, hipStream_t s);
int head_bwd_pair(const HeadBwd& a, const HeadBwd& e, const HeadLossFin& lf, hipStream_t s);
// cnt (optional): three slots that receive B as a double (row counts of the head's BatchNorms, summed with the statistics)
int head_colstats(const float* g, int B, int F, double* st, hipStream_t s, double* cnt0 = nullptr, double* cnt1 = nullptr,
                  double* cnt2 = nullptr);
int head_fwd(const HeadFwd& a, hipStream_t s);
int head_bwd(const HeadBwd& a, hipStream_t s);
int head_gbn_bwd(const HeadGbn& a, hipStream_t s);

// The final reduction of a layer's edge gradients (8 shared accumulator slabs -> d att.weight, d self_r) handed to the NEXT
// layer's first backward kernel instead of a launch of its own (unpack_grads has nothing else to do for a layer on the balanced
// GEMM): eacc == nullptr means nothing is pending.
struct EdgeDrain {
    double* eacc; int K;
    float* datt_w[EAGCN_MAX_VIEWS]; float* dself_r[EAGCN_MAX_VIEWS]; int channels[EAGCN_MAX_VIEWS];
    const float* rsig;           // [K] sigmoid(self_r) of the layer the gradients belong to
};
// dpad_views: dpad_row holds one gradient row per VIEW ([K][ld_out]: sampled dropout of the non-stored rows) instead of one
int layer_backward_impl(const eagcn_batch* b, const eagcn_layer_params* p, const eagcn_layer_bufs* w,
                        const float* dxout, const ReadoutGrad* rg, const float* dpad_row, float* dx,
                        const eagcn_layer_grads* g, void* stream, bool dpad_views = false,
                        const ZeroJob* zero_after = nullptr, const EdgeDrain* drain_in = nullptr, EdgeDrain* drain_out = nullptr);
// skip_apply: stop after the BatchNorm table (the caller applies it while it consumes Y: fused read-out of the top layer)
// planes_only: the output leaves as the operand planes of the next layer's products ONLY (w->xout_planes; the fp32 matrix w->xout
// is not written): the model engine asks for it when layer_reads_planes_only() holds for the layer above
int layer_forward_impl(const eagcn_batch* b, const eagcn_layer_params* p, const eagcn_layer_bufs* w, void* stream,
                       bool prepacked, bool skip_apply = false, bool planes_only = false);
// true when BOTH directions of layer p take their input x from its bf16 plane images and never from the fp32 matrix (forward
// product and the dX / dW pair on gemm_bx3.hip: the same conditions layer_forward_impl / layer_backward_impl test); a layer that
// is then handed w->x == nullptr fails loudly if it reaches a fallback
bool layer_reads_planes_only(const eagcn_batch* b, const eagcn_layer_params* p, bool aux_stream);
// relu / dropout / mask / view merge of a layer from its saved Y and BatchNorm table (what layer_forward_impl ends with)
int layer_apply_impl(const eagcn_batch* b, const eagcn_layer_params* p, const eagcn_layer_bufs* w, void* stream);
int pack_params_all(const eagcn_batch* b, const eagcn_layer_params* const* ps, void* const* packed,
                    const size_t* packed_bytes, int n, void* stream, const ZeroJob* zj = nullptr, const HandOff* ho = nullptr);
struct ReadoutBn {
    const float* Y; int ldy;                 // [T][Fp] pre-BatchNorm (bias-free) aggregation output
    const float* bn; int fp;                 // [4][Fp] scale / shift / ...
    int do_drop; uint32_t thr; float inv_keep; uint64_t seed; const uint64_t* seed_dev;
    const int64_t* size; int mode;
    float* g; int F;                         // [B][F]
    double* st;                              // [2 F] sum g, sum g^2 (fp64 atomics; zero on entry)
    int st_copies = 1, st_stride = 0;        // replicas of st (HeadFwd): workgroup x adds to replica x % st_copies
    bool pair = false;                       // set by the launcher: a lane quad shares a dropout draw
    double *cnt0, *cnt1, *cnt2;              // optional row-count slots of the head's BatchNorms (sync-BatchNorm)
};
int readout_bn_forward(const eagcn_batch* b, const eagcn_layout* lay, const ReadoutBn& a, void* stream);
int readout_forward_sampled(const eagcn_batch* b, const float* x, const eagcn_layout* lay, const eagcn_layer_params* p,
                            const float* bn_sh, const int64_t* size, int mode, float* g, int F, uint16_t* cnt, float* padc,
                            uint32_t* tab, void* stream);   // tab: (N + 1)^2 words of scratch
int readout_backward_pad_views(const eagcn_batch* b, const float* dg, const eagcn_layout* lay, const int64_t* size, int mode,
                               int F, int K, const uint16_t* cnt, float dropout, float* dpad, void* stream);
int readout_backward_pad(const eagcn_batch* b, const float* dg, const eagcn_layout* lay, const int64_t* size,
                         int mode, int F, float* dpad_row, void* stream);

}  // namespace eagcn
