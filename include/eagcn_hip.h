/*
 * eagcn_hip.h -- C ABI of the MI355X (gfx950) EAGCN hot path.
 *
 * Drop-in boundary.  The reference (Luckick/EAGCN) has no FFI of its own: the hot path is the
 * Python nn.Module surface of eagcn_pytorch/layers.py and models.py, whose bodies are chains of
 * ATen calls.  Each entry point below replaces one such body; the reference-side binding a
 * maintainer would add is a ctypes stub inside the module's forward (shown in INTEGRATION.md,
 * implemented in eagcn_amd/_lib.py + eagcn_amd/layers.py).
 *
 *   eagcn_index_*        <- layers.py:294-304 (masks), layers.py:82 (1x1 conv over one-hot
 *                           relation tensors == dictionary lookup), utils.py:575-640 (layout)
 *   eagcn_index_from_bonds <- the same index from a compact bond list: replaces the dense padding of
 *                           utils.py:586-640 + neural_fp.py:57-122 on the device side
 *   eagcn_pack_rows / eagcn_unpack_rows
 *                        <- the padded [B,N,F] activation layout of layers.py:293 / models.py:96
 *   eagcn_layer_forward  <- GraphConv_Layer.forward, layers.py:293-316 with
 *                           GraphConv_block.forward layers.py:81-95, GraphConv_base.forward
 *                           layers.py:38-45, AFM_BatchNorm.forward layers.py:408-412,
 *                           Ave_multi_view.forward layers.py:431-437
 *   eagcn_layer_backward <- autograd of the above (the reference has no hand-written backward)
 *   eagcn_attention_dense<- the A_weight return value, layers.py:318 (stack of A1, layers.py:83)
 *   eagcn_readout_*      <- models.py:108-111 (sum / ave over atoms)
 *   eagcn_pool_*         <- Diff_Pooling.forward layers.py:498-506 as used by models.py:104-106 (molfp_mode='pool'), with
 *                           the last layer's A_weight of layers.py:319-324 (and its autograd backward)
 *   eagcn_gemm_f32       <- Dense.forward layers.py:382-387 (x @ W), used by models.py:114-120
 *   eagcn_bce_loss / eagcn_mse_loss
 *                        <- the loss of train.py:321-331 incl. weight_tensor utils.py:653-679
 *   eagcn_model_forward / eagcn_model_backward
 *                        <- EAGCN.forward models.py:96-121 end to end (layers, read-out, Graph_BN,
 *                           den1/bn_den1/relu/dropout/den2/bn_den2/relu/den3) and its autograd backward
 *
 * Conventions: all pointers are device pointers unless named host_*; tensors are fp32,
 * contiguous, row-major; `stream` is a hipStream_t passed as void*; functions return 0 on
 * success and a negative code on failure with a message available from eagcn_last_error().
 * No function synchronises the stream.  Nothing here falls back to a CPU path.
 *
 * Packed ("compact") activation layout: molecule b occupies rows row0[b] .. row0[b]+nat[b]-1 of a
 * [T, ld] matrix, nat[b] = 1 + (largest atom index that has a bond).  Rows of the padded
 * reference tensor beyond nat[b] (padding and trailing bond-less atoms) are never stored: every
 * one of them has the same value, which the kernels account for analytically.  Feature columns
 * are grouped in segments (one per view for a Concate layer output), each segment padded with zero
 * columns to a multiple of 16.
 */
#ifndef EAGCN_HIP_H
#define EAGCN_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define EAGCN_MAX_VIEWS 8
#define EAGCN_MAX_SEGS 8
#define EAGCN_MAX_CHANNELS 255
#define EAGCN_STRUCT_CONCATE 0
#define EAGCN_STRUCT_WEIGHTED 1

#define EAGCN_OK 0
#define EAGCN_ERR_ARG (-1)
#define EAGCN_ERR_HIP (-2)
#define EAGCN_ERR_SCRATCH (-3)

/* meta[] slots written by eagcn_index_build (device) and mirrored to host_meta */
#define EAGCN_META_T 0        /* packed rows                                   */
#define EAGCN_META_NMAX 1     /* max nat[b]                                    */
#define EAGCN_META_NTILES 2   /* sum_b ceil(nat[b]/16)                         */
#define EAGCN_META_BAD_ADJ 3  /* # adjacency entries not in {0,1}              */
#define EAGCN_META_BAD_REL 4  /* # bonded (i,j,view) whose channels are not one-hot */
#define EAGCN_META_NEDGE 5    /* # directed edges                              */
#define EAGCN_META_OVERFLOW 6 /* packed rows of a batch that does not fit the row capacity eagcn_batch.T (0 if it
                                 fits): such a batch is indexed as EMPTY (meta[T] = meta[NTILES] = 0), so that no
                                 kernel touches memory beyond the capacity-sized buffers              */
#define EAGCN_META_EDGE_OVERFLOW 7 /* directed bonds of a batch that does not fit the edge capacity eagcn_batch.E (0 if it
                                 fits); handled like EAGCN_META_OVERFLOW: the batch is indexed as empty                   */
#define EAGCN_META_NLOG 8     /* logical padded molecule size of THIS batch when it is smaller than the capacity eagcn_batch.N
                                 (eagcn_batch.n_logical): the reference pads every batch to its own maximum (utils.py:583) and
                                 counts the padding rows in its BatchNorm statistics and filler weights, so kernels take
                                 B * N_logical, N_logical - nat[b] from here; 0 = eagcn_batch.N                           */
#define EAGCN_META_NBLK 9     /* row blocks of eagcn_batch.blk (bond-list aggregation, csrc/lagg.hip)               */
#define EAGCN_META_WORDS 16

typedef struct eagcn_batch {
    int32_t B, N, K;
    int32_t ldc;                            /* row stride of code maps in bytes, multiple of 16 */
    int32_t T, n_max, n_tiles;              /* T / n_tiles: CAPACITIES (row stride of [K][T] buffers,
                                               grid sizing); the actual counts are meta[T], meta[NTILES]
                                               on the device and every kernel reads them there, so the
                                               caller may pass upper bounds without a host read-back  */
    int32_t channels[EAGCN_MAX_VIEWS];
    uint8_t* code;                          /* [K][B][N][ldc] 0 = no bond, c+1 = bond type c;
                                               rows with deg_bn == 0 are undefined (never given weight) */
    int32_t* deg_bn;                        /* [B][N] degree of every padded row                */
    int32_t* nat;                           /* [B]                                              */
    int32_t* row0;                          /* [B+1] exclusive prefix of nat                    */
    int32_t* tile0;                         /* [B+1] exclusive prefix of ceil(nat/16)           */
    int32_t* meta;                          /* [EAGCN_META_WORDS]                               */
    int32_t* row_mol;                       /* [T]                                              */
    int32_t* row_loc;                       /* [T] atom index inside the molecule               */
    float* row_m;                           /* [T] row mask m_i = max_j adj[i,j]                */
    int32_t* row_deg;                       /* [T]                                              */
    int32_t* tile_mol;                      /* [n_tiles]                                        */
    int32_t* row_info;                      /* [T][4] {molecule, atom, nat[mol], row0[mol]}: one
                                               16-byte load instead of a two-hop lookup          */
    int32_t* tile_info;                     /* [n_tiles][4] {molecule, row tile, nat[mol], row0[mol]} */
    /* General (non one-hot) relation tensors, layers.py:82: at a bond the reference evaluates sigmoid(sum_c w[c] R[c,i,j]) for
       ANY channel values.  The index keeps bond-type CODES; a code's meaning is a channel VECTOR: rel_vec[k] = [channels[k]]
       [rel_c[k]] floats (code c+1 -> row c), built per batch from the distinct vectors at the bonds (eagcn_amd/collate.py; at
       most 255 per view).  NULL (the default): code c+1 means the one-hot vector e_c, i.e. sigma(w[c]).                 */
    const float* rel_vec[EAGCN_MAX_VIEWS];
    int32_t rel_c[EAGCN_MAX_VIEWS];         /* channels of view k's attention weight when rel_vec[k] is set           */
    /* Bond lists (built by eagcn_index_rows from the code maps when build_lists is set): the GAT baseline layers
       (csrc/gat.hip, layers.py:99-203) walk them -- scores per atom, softmax over the deg+1 entries of a row.       */
    int32_t E;                              /* CAPACITY of the four edge arrays (directed bonds)             */
    int32_t n_logical;                      /* HOST input of the index entry points: padded size N_in <= N of the caller's
                                               tensors for this batch (their row stride), 0 = N; mirrored to meta[NLOG] */
    int32_t* ecnt;                          /* [B]   directed bonds of each molecule                         */
    int32_t* edge0;                         /* [B+1] exclusive prefix of ecnt                                */
    int32_t* mol_info;                      /* [B][4] {nat, row0, edge0, directed bonds}: one 16-byte load   */
    int32_t* row_ptr;                       /* [T][2] {first entry, count}: bonds (i,j) of packed row i      */
    int32_t* col_ptr;                       /* [T][2] {first entry, count}: bonds (i,j) INTO packed row j    */
    int32_t* nbr;                           /* [E] row lists: atom index j inside the molecule               */
    int32_t* tnbr;                          /* [E] column lists: atom index i inside the molecule            */
    uint64_t* ecode;                        /* [E] row lists: byte k = bond-type code of view k (1-based)    */
    uint64_t* tcode;                        /* [E] column lists: the same for bond (i,j)                     */
    int32_t build_lists;                    /* HOST input of eagcn_index_rows: build the bond lists (GAT layers, the
                                               LDS-staged bond-list aggregation of csrc/lagg.hip)              */
    int32_t t_hint;                         /* HOST hint: about how many packed rows batches of this shape really hold (0: unknown ->
                                               T).  Only steers size-dependent kernel CHOICES whose launch is baked into a captured
                                               graph (the plane GEMM's tile shape); every kernel is correct for any actual count.   */
    int32_t* blk;                           /* [8 (B + 1)] (optional, 16-byte aligned; built by eagcn_index_rows with the bond lists):
                                               ROW BLOCKS of the LDS-staged aggregation (csrc/lagg.hip) -- whole molecules, greedily
                                               packed to at most 256 packed rows and 16 molecules; block q = two int4 records {first
                                               molecule, molecules, first packed row, rows} {first list entry, entries, 0, 0};
                                               meta[NBLK] blocks                                                                    */
} eagcn_batch;

/* column layout of a packed activation matrix */
typedef struct eagcn_layout {
    int32_t nseg;
    int32_t width[EAGCN_MAX_SEGS];          /* exact widths                                     */
    int32_t pad[EAGCN_MAX_SEGS];            /* padded widths (sum = ld)                         */
} eagcn_layout;

typedef struct eagcn_layer_params {
    int32_t K;                              /* views                                            */
    int32_t structure;                      /* EAGCN_STRUCT_*                                   */
    int32_t training;                       /* BatchNorm batch statistics + dropout             */
    int32_t width[EAGCN_MAX_VIEWS];         /* F_k                                              */
    eagcn_layout in;                        /* layout of x                                      */
    float dropout;
    float bn_eps, bn_momentum;
    uint64_t seed;                          /* dropout stream for this call                     */
    const uint64_t* seed_dev;               /* if non-NULL the seed is read from this DEVICE word  */
                                            /* instead (a replayed HIP graph cannot change by-value */
                                            /* arguments)                                          */
    const float* att_w[EAGCN_MAX_VIEWS];    /* [C_k]    blockK.att.weight                       */
    const float* self_r[EAGCN_MAX_VIEWS];   /* [1]      blockK.self_r                           */
    const float* W[EAGCN_MAX_VIEWS];        /* [fin,F_k] blockK.graph_conv.weight               */
    const float* bias[EAGCN_MAX_VIEWS];     /* [F_k]    blockK.graph_conv.bias                  */
    const float* gamma[EAGCN_MAX_VIEWS];    /* [F_k]    blockK.batch_norm.bn.weight             */
    const float* beta[EAGCN_MAX_VIEWS];     /* [F_k]    blockK.batch_norm.bn.bias               */
    float* run_mean[EAGCN_MAX_VIEWS];       /* [F_k]    updated in place when training          */
    float* run_var[EAGCN_MAX_VIEWS];
    const float* ave_w;                     /* [K]      layer.ave.weight (Weighted_sum)         */
} eagcn_layer_params;

/* Cross-rank sum of `n` doubles in place on `stream` (sync-BatchNorm, SURVEY.md 8e "BN modes (ii)"): the per-channel partial
 * sums of a layer's BatchNorm (sum y, sum y^2, rows -- backward: sum dH, sum dH xhat, rows) are handed to the caller between
 * the reduction and the finalize kernel of that BatchNorm, forward and backward; the caller all-reduces them (RCCL through
 * torch.distributed in eagcn_amd/parallel.py -- also while the call is being captured into a HIP graph, the collective then
 * becomes a node of that graph).  Returns 0 on success.  NULL hook: BatchNorm over this rank's rows only ("local-BN"). */
typedef int (*eagcn_allreduce_fn)(double* buf, int n, void* stream, void* user);

typedef struct eagcn_layer_bufs {
    const float* x;                         /* [T][ld_in]                                       */
    float* P;                               /* [T][Fp]  X.[W_1|..|W_K]            (saved)       */
    float* Y;                               /* [T][Fp]  A_k.P_k, pre-bias, pre-BN (saved)       */
    float* rscale;                          /* [K][T]   m_i / rowsum_i            (saved)       */
    float* bn;                              /* [4][Fp]  scale, shift, mean, invstd (saved)      */
    float* xout;                            /* [T][ld_out]                                      */
    float* pad_row;                         /* [ld_out] value of every non-stored row of xout   */
    void* scratch;
    size_t scratch_bytes;
    void* aux_stream;                       /* optional second hipStream_t: backward runs the edge  */
                                            /* gradients and the dW product there, concurrently     */
                                            /* with the dX chain (fork/join by events; capturable)  */
    void* packed;                           /* optional, eagcn_layer_packed_bytes(): forward keeps  */
    size_t packed_bytes;                    /* the re-laid parameters here and backward reuses them */
    eagcn_allreduce_fn stats_hook;          /* sync-BatchNorm: cross-rank sum of the BatchNorm partial sums (NULL: local-BN) */
    void* stats_user;
    /* bf16 plane images (gemm mode 3 / 4, csrc/gemm_bx3.hip: the layer products run on the bf16 matrix cores from operands    */
    /* their producers already split): x_planes = the plane images of x (panel-major images of row capacity eagcn_batch.T and   */
    /* ld_in columns, plane stride eagcn_bx3_plane_elems(T, ld_in) -- layout below at eagcn_bx3_split; three planes in mode 3,   */
    /* one in mode 4), written by the layer below through ITS xout_planes; NULL: the layer splits x itself (one more launch) /   */
    /* does not write them.  Ignored in the fp32 modes and for layers narrower than 128 input columns.                          */
    const uint16_t* x_planes;
    uint16_t* xout_planes;                  /* [planes] images of [T rows][ld_out], stride eagcn_bx3_plane_elems(T, ld_out)    */
} eagcn_layer_bufs;

typedef struct eagcn_layer_grads {
    float* dW[EAGCN_MAX_VIEWS];
    float* dbias[EAGCN_MAX_VIEWS];
    float* dgamma[EAGCN_MAX_VIEWS];
    float* dbeta[EAGCN_MAX_VIEWS];
    float* datt_w[EAGCN_MAX_VIEWS];
    float* dself_r[EAGCN_MAX_VIEWS];
    float* dave_w;                          /* [K] or NULL                                      */
} eagcn_layer_grads;

/* 1 when batches of this shape take a bond-list form of the aggregation in either direction (csrc/lagg.hip: a block of up to 256
 * packed rows staged in LDS, each row gathers its bonded rows, one rank-one term per molecule -- instead of the dense nat x nat
 * block on the matrix cores; default for large molecules, for batches of up to 256 molecules and for the backward of Concate
 * layers, padded sizes up to 256 atoms): the index must then carry bond lists and row
 * blocks -- set eagcn_batch.build_lists = 1 (and eagcn_batch.blk) before eagcn_index_rows.  `structure`: EAGCN_STRUCT_* of the
 * layers the index will serve, -1 when not known (the two-argument form). */
int eagcn_agg_wants_bond_lists(int B, int N);
int eagcn_agg_wants_bond_lists_for(int B, int N, int structure);

/* ---- library ------------------------------------------------------------------------------- */
int eagcn_abi_version(void);           /* 7 (round 6, second half: eagcn_head_*); bumped with every struct-layout / signature change */
size_t eagcn_struct_size(int which);   /* 0 batch, 1 layout, 2 layer_params, 3 layer_bufs, 4 layer_grads,
                                          5 head_params, 6 head_grads, 7 model, 8 gat_params, 9 pool_att */
const char* eagcn_last_error(void);
int eagcn_pad16(int width);
int eagcn_layer_out_ld(const eagcn_layer_params* p);   /* ld of xout                            */
int eagcn_layer_fp(const eagcn_layer_params* p);       /* Fp = sum_k pad16(F_k)                 */
size_t eagcn_layer_packed_bytes(const eagcn_batch* b, const eagcn_layer_params* p);
size_t eagcn_layer_fwd_scratch_bytes(const eagcn_batch* b, const eagcn_layer_params* p);
size_t eagcn_layer_bwd_scratch_bytes(const eagcn_batch* b, const eagcn_layer_params* p);

/* ---- batch index ----------------------------------------------------------------------------- */
/* stage 1: needs code, deg_bn, nat, row0, tile0, meta; copies meta to host_meta (pinned) async */
int eagcn_index_build(const float* adj, const float* const* rel, eagcn_batch* b,
                      int32_t* host_meta, void* stream);
/* stage 1 from a COMPACT batch instead of the dense collate tensors: E directed bonds given as
 * bond_mol/bond_i/bond_j [E] (int32) and bond_code [E][K] (uint8, type index per view, < channels[k]).
 * Produces exactly the index eagcn_index_build derives from adj + one-hot relation tensors; touches
 * O(E) input bytes instead of 4*(1+sum C_k)*B*N*N. */
int eagcn_index_from_bonds(const int32_t* bond_mol, const int32_t* bond_i, const int32_t* bond_j,
                           const uint8_t* bond_code, int64_t E, eagcn_batch* b, int32_t* host_meta, void* stream);
/* stage 2 (after the caller read host_meta and allocated the per-row arrays) */
int eagcn_index_rows(const eagcn_batch* b, void* stream);

/* ---- layout conversion ----------------------------------------------------------------------- */
int eagcn_pack_rows(const eagcn_batch* b, const float* dense, int F, const eagcn_layout* lay,
                    float* packed, void* stream);
/* The GAT baseline layer (reference layers.py:99-203, models.py:69-73): x [T][ld_in] packed rows with a
   single-segment layout -> xout [T][pad16(F)].  h [T][pad16(F)] and s12 [2][T] are saved for the backward.  Needs a batch
   index with bond lists (eagcn_batch.build_lists).  Attention dropout 0.5 (layers.py:104) and the layer dropout are applied in
   training mode from the counter-based stream `seed`.                                                                      */
typedef struct eagcn_gat_params {
    int32_t fin, ld_in, F, training;
    float alpha, att_dropout, dropout;      /* leaky-relu slope (0.2), attention dropout (0.5), layer dropout          */
    float reserved_;
    uint64_t seed;
    const float* W;                         /* [fin][F]  graph_conv.W                                                   */
    const float* a;                         /* [2F]      graph_conv.a                                                   */
    const uint64_t* seed_dev;               /* optional: the seed is read from device memory (captured launches)       */
} eagcn_gat_params;
size_t eagcn_gat_scratch_bytes(const eagcn_batch* b, int F);
int eagcn_gat_forward(const eagcn_batch* b, const eagcn_gat_params* p, const float* x, float* h, float* s12, float* xout,
                      void* stream);
int eagcn_gat_backward(const eagcn_batch* b, const eagcn_gat_params* p, const float* x, const float* h, const float* s12,
                       const float* xout, const float* dxout, float* dx, float* dW, float* da, void* scratch,
                       size_t scratch_bytes, void* stream);

/* ---- Diff_Pooling read-out, molfp_mode='pool' (layers.py:492-506, models.py:90-92, 104-106) ------------------------------------
   g[b] = sum_p relu(S^T . relu((A.x).Wf)),  S = softmax((A.x).Ws), A = the attention matrix the LAST layer returns.
   The call sequence of one forward: eagcn_pool_attention_forward (A as packed rows [T][lda], lda >= N; rinv [T] = 1 / row sum;
   padsum [T] = total weight of the non-stored columns) -> eagcn_pool_mix_forward (AX [T][F] = A.x, exact columns) ->
   eagcn_gemm_f32 (Z [T][ldz] = AX . [Wf | Ws], F + P columns) -> eagcn_pool_reduce_forward (S [T][P], Pm [B][P][F], g [B][F]).
   Backward: the same entry points mirrored (reduce -> two GEMMs -> mix -> attention).  The pooled adjacency S^T.A.S that
   layers.py:504 also returns is never read by models.py and is not computed.                                             */
#define EAGCN_POOL_MAX 8                    /* clusters P (pool_num, models.py:25: 5)                                        */
typedef struct eagcn_pool_att {
    int32_t K;                              /* views of the layer that produced A (1 for the baselines)                      */
    int32_t mode;                           /* 0: edge-attention layer with last=True (layers.py:319-324)
                                               1: Vanilla_GCN (layers.py:250-253)   2: GAT (layers.py:189: adj + mask*I)     */
    int32_t att_c[EAGCN_MAX_VIEWS];         /* entries of att_w[k] / datt_w[k]                                               */
    const float* att_w[EAGCN_MAX_VIEWS];    /* mode 0: blockK.att.weight                                                     */
    const float* ave_a;                     /* mode 0: [K] ave_A.weight                                                      */
    const float* self_r;                    /* mode 0: [1] the LAYER's self_r (layers.py:287)                                */
    float* datt_w[EAGCN_MAX_VIEWS];         /* backward outputs (mode 0); NULL entries are skipped                           */
    float* dave_a;
    float* dself_r;
} eagcn_pool_att;
size_t eagcn_pool_scratch_bytes(void);
int eagcn_pool_attention_forward(const eagcn_batch* b, const eagcn_pool_att* p, float* A, int lda, float* rinv, float* padsum,
                                 void* stream);
int eagcn_pool_attention_backward(const eagcn_batch* b, const eagcn_pool_att* p, const float* A, int lda, const float* rinv,
                                  const float* dA, void* scratch, size_t scratch_bytes, void* stream);
/* x: packed activations [T][ld(lay)] of the last layer, pad_row [ld] (or NULL = 0): the value of every non-stored row */
int eagcn_pool_mix_forward(const eagcn_batch* b, const eagcn_layout* lay, const float* A, int lda, const float* padsum,
                           const float* x, const float* pad_row, float* AX, int F, void* stream);
/* dA [T][lda], dx [T][ld] (ZERO-FILLED by the caller: only exact columns are written), dpad_row [ld]; each may be NULL */
int eagcn_pool_mix_backward(const eagcn_batch* b, const eagcn_layout* lay, const float* A, int lda, const float* padsum,
                            const float* x, const float* pad_row, const float* dAX, int F, float* dA, float* dx,
                            float* dpad_row, void* stream);
int eagcn_pool_reduce_forward(const eagcn_batch* b, const float* Z, int ldz, int F, int P, float* S, float* Pm, float* g,
                              void* stream);
int eagcn_pool_reduce_backward(const eagcn_batch* b, const float* Z, int ldz, int F, int P, const float* S, const float* Pm,
                               const float* dg, float* dZ, void* stream);

/* Device half of the reference's collate (utils.py:504-640: every molecule zero-padded to the batch maximum): per-molecule
   atom-feature rows, concatenated [sum n_b][F] with molecule b = rows mol_offset[b] .. mol_offset[b+1], -> padded [B][N][F].
   Together with eagcn_index_from_bonds a batch reaches the device as O(atoms + bonds) bytes instead of
   4 (1 + sum C_k) B N^2.                                                                     */
int eagcn_pad_rows(const float* rows, const int32_t* mol_offset, int B, int N, int F, float* out, void* stream);

int eagcn_unpack_rows(const eagcn_batch* b, const float* packed, const eagcn_layout* lay,
                      const float* pad_row, float* dense, int F, void* stream);

/* ---- graph-conv layer ------------------------------------------------------------------------ */
int eagcn_layer_forward(const eagcn_batch* b, const eagcn_layer_params* p,
                        const eagcn_layer_bufs* w, void* stream);
/* dxout: [T][ld_out]; dpad_row: [ld_out] gradient w.r.t. bufs.pad_row (the common value of all
 * non-stored rows; Weighted_sum only) or NULL; dx: [T][ld_in] or NULL */
int eagcn_layer_backward(const eagcn_batch* b, const eagcn_layer_params* p,
                         const eagcn_layer_bufs* w, const float* dxout, const float* dpad_row,
                         float* dx, const eagcn_layer_grads* g, void* stream);
/* A1_k = sigmoid(w_k[type]) * adj for every view: out [K][B][N][N] */
int eagcn_attention_dense(const eagcn_batch* b, const eagcn_layer_params* p, float* out,
                          void* stream);

/* ---- read-out -------------------------------------------------------------------------------- */
/* g[b][f] = sum_i x[b,i,f] (all N rows; non-stored rows contribute pad_row) ; mode 1: / size[b] */
int eagcn_readout_forward(const eagcn_batch* b, const float* x, const eagcn_layout* lay,
                          const float* pad_row, const int64_t* size, int mode, float* g, int F,
                          void* stream);
/* dx: [T][ld]; dpad_row: [ld] or NULL */
int eagcn_readout_backward(const eagcn_batch* b, const float* dg, const eagcn_layout* lay,
                           const int64_t* size, int mode, int F, float* dx, float* dpad_row,
                           void* stream);

/* ---- whole model: one call forward, one call backward (reference models.py:96-121) -------------- */
typedef struct eagcn_head_params {
    int32_t f_in, n_den1, n_den2, nclass;
    float dropout, bn_eps, bn_momentum;
    const float *den1_w, *den2_w, *den3_w;          /* [f_in,n_den1] [n_den1,n_den2] [n_den2,nclass]   */
    const float *gbn_w, *gbn_b;  float *gbn_rm, *gbn_rv;   /* Graph_BN   (models.py:80-88, 112)        */
    const float *bn1_w, *bn1_b;  float *bn1_rm, *bn1_rv;   /* bn_den1    (models.py:115)               */
    const float *bn2_w, *bn2_b;  float *bn2_rm, *bn2_rv;   /* bn_den2    (models.py:119)               */
} eagcn_head_params;

typedef struct eagcn_head_grads {
    float *d_den1_w, *d_den2_w, *d_den3_w, *d_gbn_w, *d_gbn_b, *d_bn1_w, *d_bn1_b, *d_bn2_w, *d_bn2_b;
} eagcn_head_grads;

typedef struct eagcn_model {
    int32_t n_layers;                       /* 1..4                                                    */
    int32_t molfp_mode;                     /* 0 = 'sum', 1 = 'ave' (models.py:108-111)                */
    int32_t training;
    uint64_t head_seed;                     /* dropout stream of the head (models.py:116)              */
    const uint64_t* head_seed_dev;          /* device-resident alternative (see eagcn_layer_params)    */
    int32_t input_packed;                   /* 1: saved already holds the packed input (eagcn_model_pack_input) */
    void* aux_stream;                       /* optional second stream for off-critical-path backward work */
    eagcn_layer_params layer[4];            /* layer[l].in must equal the output layout of layer l-1   */
    eagcn_head_params head;
    eagcn_allreduce_fn stats_hook;          /* sync-BatchNorm of EVERY BatchNorm of the model (the per-view ones of the layers and
                                               Graph_BN / bn_den1 / bn_den2 of the head; see eagcn_layer_bufs); NULL: local-BN   */
    void* stats_user;
    int32_t stats_world;                    /* ranks behind the hook (>= 1): the head's d gamma / d beta are formed from the
                                               summed sums and divided by it, so that the gradient AVERAGE over ranks is exact */
    int32_t fuse_readout;                   /* 1: a Concate top layer does not materialise its output matrix in the forward: its
                                               relu / dropout / mask are applied while the read-out sums the atoms (one launch
                                               instead of bn_apply + read-out + column statistics).  The matrix
                                               (atom_representations, models.py:102) is built on request by
                                               eagcn_model_atom_rep_materialize.  Ignored for other structures.            */
    uint32_t* fwd_signal;                   /* optional device word: the forward adds 1 to it behind the read-out (the head's first
                                               launch does, with a relaxed atomic: the word places work, it guards no data), i.e. when the
                                               layer products / aggregations of the step have been issued and only short kernels
                                               follow for a while (head, loss, head backward, the top BatchNorm backward).  Another
                                               stream can wait for a count with eagcn_stream_wait_counter: the caller's batch
                                               preparation for the NEXT step is placed under those kernels instead of beside the
                                               persistent GEMMs (eagcn_amd/graph.py).  NULL: no signal                        */
    uint32_t* wait_flag;                    /* optional device word: the forward's FIRST launch polls it until it is non-zero and
                                               clears it -- the batch-ready hand-off from the stream that built the index and packed
                                               the input (eagcn_stream_signal_flag behind its last kernel) without a
                                               hipStreamWaitEvent, whose cross-queue dependency costs the waiting stream ~40 us per
                                               step on ROCm 7.2.  The signal must already be enqueued when the forward is issued
                                               (on a stream that runs concurrently with this one).  NULL: no wait               */
    uint32_t* start_signal;                 /* optional device word: the forward's FIRST launch adds 1 to it -- everything issued on the
                                               stream before this forward (the previous steps: their index / saved blocks are free) has
                                               completed.  The stream that prepares the batch after next waits for the count
                                               (eagcn_stream_wait_counter) instead of for a stream event recorded between two step
                                               graphs (recording one that another stream waits for costs the recording stream ~27 us
                                               per step on ROCm 7.2).  NULL: no signal.  Both hand-offs ride in the parameter-packing
                                               launch (no launch of their own) when the input arrives packed (input_packed = 1)       */
} eagcn_model;

/* *flag = 1 behind everything issued on `stream` so far (release, device scope) */
int eagcn_stream_signal_flag(uint32_t* flag, void* stream);
/* one parked wavefront on `stream` that polls *flag until it is non-zero, then clears it (what eagcn_model.wait_flag does in front of a
 * forward); after budget_seconds without a signal it gives up and raises the sticky word below */
int eagcn_stream_wait_flag(uint32_t* flag, double budget_seconds, void* stream);
int eagcn_stream_wait_timeouts(void);   /* non-zero once any flag wait gave up (host-mapped word, no synchronisation) */
void eagcn_stream_wait_reset(void);

/* `stream` does not run anything issued behind this call until *counter (device memory, see eagcn_model.fwd_signal / start_signal)
 * >= value: one parked wavefront that polls the word.  The signalling work must already be enqueued (or be enqueued without
 * waiting for this stream); after 2 s without it the poll gives up and raises the sticky word of eagcn_stream_wait_timeouts. */
int eagcn_stream_wait_counter(const uint32_t* counter, uint32_t value, void* stream);

size_t eagcn_model_saved_bytes(const eagcn_batch* b, const eagcn_model* m);    /* kept forward -> backward */
size_t eagcn_model_scratch_bytes(const eagcn_batch* b, const eagcn_model* m);  /* transient, either call  */
/* where the last layer's packed activations [T][ld] and pad_row [ld] live inside `saved` */
int eagcn_model_atom_rep(const eagcn_batch* b, const eagcn_model* m, size_t* xout_offset,
                         size_t* pad_row_offset, int* ld);
/* builds the last layer's packed output [T][ld] (+ pad_row) inside `saved` from the saved pre-BatchNorm matrix when the forward
 * ran with fuse_readout = 1 (same values as a forward without the fusion, same dropout masks); a no-op otherwise */
int eagcn_model_atom_rep_materialize(const eagcn_batch* b, const eagcn_model* m, void* saved, size_t saved_bytes, void* stream);
/* packs afm [B][N][n_afeat] into the input slot of `saved` (the first step of eagcn_model_forward) */
int eagcn_model_pack_input(const eagcn_batch* b, const eagcn_model* m, const float* afm, void* saved,
                           size_t saved_bytes, void* stream);
/* afm: dense [B][N][n_afeat]; out: [B][nclass]; graph_rep: [B][n_den2] (den2 output, models.py:118) */
int eagcn_model_forward(const eagcn_batch* b, const eagcn_model* m, const float* afm, const int64_t* size,
                        void* saved, size_t saved_bytes, void* scratch, size_t scratch_bytes, float* out,
                        float* graph_rep, void* stream);
/* dgraph_rep may be NULL; lg: n_layers entries */
int eagcn_model_backward(const eagcn_batch* b, const eagcn_model* m, const int64_t* size, void* saved,
                         size_t saved_bytes, void* scratch, size_t scratch_bytes, const float* dout,
                         const float* dgraph_rep, const eagcn_layer_grads* lg, const eagcn_head_grads* hg,
                         void* stream);

/* The same backward in two pieces, so that a data-parallel caller can start the gradient all-reduce of the upper layers while the
 * lower ones are still running: with_head != 0 runs the head + read-out backward first; then the layers layer_hi, layer_hi-1,
 * ..., layer_lo (n_layers-1 >= layer_hi >= layer_lo >= 0).  eagcn_model_backward == (with_head = 1, n_layers-1, 0); a split
 * must be issued top-down on one stream: (1, n_layers-1, l) then (0, l-1, 0). */
int eagcn_model_backward_range(const eagcn_batch* b, const eagcn_model* m, const int64_t* size, void* saved,
                               size_t saved_bytes, void* scratch, size_t scratch_bytes, const float* dout,
                               const float* dgraph_rep, const eagcn_layer_grads* lg, const eagcn_head_grads* hg,
                               int with_head, int layer_hi, int layer_lo, void* stream);

/* ---- a training step's forward + loss + HEAD backward as one call (reference train.py:317-331 in front of models.py:112-120 and
 * their autograd): the model forward as eagcn_model_forward, then the loss on `out` and the backward of den3 / bn_den2 / den2 /
 * bn_den1 / den1 / Graph_BN.  Where a row block of the logits is one tile (nclass <= 64) and no sync-BatchNorm hook sits between the
 * stages, the last forward product, the loss and dense 3's d(input) product run as ONE launch per 16-row block (csrc/head2.hip
 * head_mid_kernel) and dense 3's weight gradient rides in dense 2's backward launch: six launches instead of eight; otherwise
 * the separate launches -- same arithmetic, same results.  On return *loss->loss, loss->dout, every head gradient
 * of `hg` and the gradient of the molecule fingerprints (inside `scratch`) exist: the caller continues with
 * eagcn_model_backward_range(..., dout = loss->dout, with_head = 0, n_layers - 1, layer_lo). */
typedef struct eagcn_step_loss {
    int32_t kind;                /* 0: weighted BCE with logits over the labelled entries (train.py:326-331, eagcn_bce_loss)
                                    1: mean squared error (train.py:321-325, eagcn_mse_loss)                                 */
    const float* labels;         /* [B][nclass]; kind 0: 1 / 0 / anything else = missing                                     */
    const float* class_weight;   /* kind 0: [nclass][2] = {w_pos, w_neg}; kind 1: unused                                     */
    float* loss;                 /* device scalar (out)                                                                      */
    const float* scale;          /* optional device scalar: loss and d loss / d out are multiplied by it (the data-parallel
                                    normalisation of eagcn_amd/parallel.py); NULL = 1                                        */
    float* dout;                 /* [B][nclass] (out): d loss / d out                                                        */
} eagcn_step_loss;
int eagcn_model_forward_step(const eagcn_batch* b, const eagcn_model* m, const float* afm, const int64_t* size,
                             void* saved, size_t saved_bytes, void* scratch, size_t scratch_bytes, float* out,
                             float* graph_rep, const eagcn_step_loss* loss, const float* dgraph_rep,
                             const eagcn_head_grads* hg, void* stream);

/* ---- the head ALONE (models.py:112-120: Graph_BN -> den1 -> bn_den1 -> relu -> dropout -> den2 = graph_representation ->
 * bn_den2 -> relu -> den3) for callers that form the molecule fingerprints g [B][f_in] themselves: the GAT baseline
 * (models.py:69-73) and the Diff_Pooling read-out (models.py:104-106), whose layers run through the layer-level entry points.
 * Same kernels, same launch sequence as the head inside eagcn_model_forward / eagcn_model_backward (four launches + one clear
 * per direction).  saved: eagcn_head_saved_bytes, written by the forward, read by the backward (the caller keeps g as well);
 * scratch: eagcn_head_scratch_bytes, transient per call.  training != 0 updates the running statistics in place and needs B > 1.
 * seed / seed_dev: the head's dropout stream (seed_dev: a device-resident seed, read by the kernels -- captured launches).
 * The backward writes d loss / d g into dg [B][f_in] and every head gradient through hg. */
size_t eagcn_head_saved_bytes(const eagcn_head_params* h, int B);
size_t eagcn_head_scratch_bytes(const eagcn_head_params* h, int B);
int eagcn_head_forward(const eagcn_head_params* h, int B, int training, uint64_t seed, const uint64_t* seed_dev, const float* g,
                       void* saved, size_t saved_bytes, void* scratch, size_t scratch_bytes, float* out, float* graph_rep,
                       void* stream);
int eagcn_head_backward(const eagcn_head_params* h, int B, int training, uint64_t seed, const uint64_t* seed_dev, const float* g,
                        void* saved, size_t saved_bytes, void* scratch, size_t scratch_bytes, const float* dout,
                        const float* dgraph_rep, const eagcn_head_grads* hg, float* dg, void* stream);

/* ---- losses of the training loop (train.py:321-331), value + d/dlogits in one launch --------------- */
/* labels [B][T] with 1 / 0 / anything else = missing; class_weight [T][2] = {w_pos, w_neg} (utils.py:681-700) */
int eagcn_bce_loss(const float* logits, const float* labels, const float* class_weight, int B, int T,
                   float* loss, float* dlogits, void* stream);
int eagcn_mse_loss(const float* pred, const float* target, int n, float* loss, float* dpred, void* stream);

/* ---- optimizer step (train.py:303 `optim.Adam(..., weight_decay=wd)`, train.py:334) as ONE launch over flat fp32 buffers: every
 * hot-path parameter in one buffer, gradients / first / second moments in buffers of the same layout (n floats, a multiple of 4,
 * 16-byte aligned).  torch.optim.Adam arithmetic.  hyper_dev = {lr, beta1, beta2, eps, weight_decay} as DOUBLES (torch's scalars
 * are Python doubles: 1 - beta2 formed from an fp32 0.999 is off by 1.7e-5) and the step count *step_dev (advanced by the launch)
 * live in device memory, so the launch can sit in a captured graph; *ticket_dev must be 0 on entry (the launch leaves it 0).
 * ticket_dev == NULL: the range is updated and the count NOT advanced (several ranges of one step: the last call brings the ticket). */
int eagcn_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, const double* hyper_dev,
                    int64_t* step_dev, uint32_t* ticket_dev, void* stream);

/* ---- evaluation outputs (train.py:130-211): append one batch to device-resident [cap][T] buffers at row `row_offset`:
 * scores = sigmoid(logits) (classification = 1) or the predictions (0), targets = labels, valid = label in {0,1} / 1 ---- */
int eagcn_eval_append(const float* logits, const float* labels, int B, int T, int classification, float* scores,
                      float* targets, uint8_t* valid, int64_t row_offset, void* stream);

/* Matrix-core path of the layer products (the reference's torch.mm and its two autograd products, layers.py:40):
 *   0 = fp32 MFMA: exact fp32 products, a k-ordered fmaf chain (csrc/gemm3.hip);
 *   1 = every fp32 operand split into three bf16 pieces by the CONSUMER on its way into LDS, six bf16 MFMA products
 *       accumulated in fp32 (csrc/gemm_x6.h; kept for comparison);
 *   2 = operands rounded to bf16 by the consumer, ONE product (csrc/gemm_x6.h; NOT the parity path);
 *   3 = fp32-equivalent products at the bf16 matrix rate (exact operand split, piece products to 3 x 2^-24 |a b|, fp32 accumulate): the PRODUCERS of a matrix (BatchNorm apply, transposed aggregation,
 *       parameter packing) write it as three bf16 planes, the products run from them by LDS-DMA (csrc/gemm_bx3.hip);
 *   4 = the same kernels on ONE bf16 plane (operands rounded to bf16, fp32 accumulation): BASELINE configs[1] "bf16".
 * The mode is read when a model's buffers are sized: set it before the first forward.  Returns the previous mode. */
int eagcn_set_gemm_mode(int mode);

/* ---- the plane GEMM of modes 3 / 4 as stand-alone entry points (tests, tools/bx3_bench.cpp) -----------------------------
 * PLANE IMAGE of a matrix with `rows_cap` rows and ld columns (ld a multiple of 8): the columns are cut into panels of 32, a panel
 * holds the rows_cap rows of its 32 columns at 64 bytes per row, and the four 16-byte chunks of a row are XOR-swizzled by the row:
 *     element (r, c) at  ((c >> 5) * rows_cap + r) * 32 + ((((c >> 3) & 3) ^ ((r >> 2) & 3)) << 3) + (c & 7)      [bf16 elements]
 * (the layout the GEMM's LDS-DMA copies verbatim: every 1 KB it moves is contiguous); eagcn_bx3_plane_elems(rows_cap, ld) elements.
 * eagcn_bx3_split writes the images of a row-major fp32 matrix [rows][ld], rows <= rows_cap: np = 3: x = x0 + x1 + x2 exactly (each
 * piece the bf16 nearest to what the pieces before it left over), np = 1: round to nearest even; plane q at planes + q *
 * plane_stride (elements, >= eagcn_bx3_plane_elems, a multiple of 8); planes 16-byte aligned. */
size_t eagcn_bx3_plane_elems(int rows_cap, int ld);
int eagcn_bx3_split(const float* x, int rows, int ld, uint16_t* planes, size_t plane_stride, int rows_cap, int np, void* stream);
/* tn = 0: C[M,N] = A[M,K].B[N,K]^T (images of A [a_rows >= M][lda >= K] and B [b_rows >= N][ldb >= K], K a multiple of 8);
 * tn = 1: C[M,N] = A[K,M]^T.B[K,N] (images [a_rows >= K][lda >= M], [b_rows >= K][ldb >= N]) as k-chunk slabs C + z * slab,
 *         z < eagcn_bx3_used_splits(splits, M, N, K) (at least 768 and at most 4096 rows of K per chunk; slabs beyond that count are
 *         NOT written), whose sum is the product. */
int eagcn_bx3_used_splits(int splits, int M, int N, int K);
/* Two kernels serve these entry points (csrc/bx3.h): 128 x 128 tiles, 4 compute + 2 loader waves (launches of about one wave of tiles)
 * and 256 x 128 tiles, 8 compute waves = two per SIMD (csrc/gemm_bx3w.hip; launches with several waves of tiles).  mode -1: picked per
 * launch by its size (default; EAGCN_BX3_WIDE in the environment presets it) | 0: always the first | 1: always the second.
 * Returns the previous setting.  The used-splits queries follow the setting. */
int eagcn_set_bx3_wide(int mode);
/* ... of the TN problem (M, N, K) of eagcn_gemm_bx3_pair: its chunks are sized against the NT problem (M0, N0, K0) of the launch */
int eagcn_bx3_pair_used_splits(int splits, int M, int N, int K, int M0, int N0, int K0);
int eagcn_gemm_bx3(int tn, int M, int N, int K, const uint16_t* A, size_t a_pstride, int lda, int a_rows, const uint16_t* B,
                   size_t b_pstride, int ldb, int b_rows, float* C, int ldc, int splits, size_t slab, int np, void* stream);
/* an NT product and a TN product (the dX / dW pair of a layer's backward) in ONE persistent launch */
int eagcn_gemm_bx3_pair(int M0, int N0, int K0, const uint16_t* A0, size_t a0_pstride, int lda0, int a0_rows, const uint16_t* B0,
                        size_t b0_pstride, int ldb0, int b0_rows, float* C0, int ldc0, int M1, int N1, int K1, const uint16_t* A1,
                        size_t a1_pstride, int lda1, int a1_rows, const uint16_t* B1, size_t b1_pstride, int ldb1, int b1_rows,
                        float* C1, int ldc1, int splits, size_t slab, int np, void* stream);

/* ---- plain fp32 MFMA GEMM (head / tests) ------------------------------------------------------ */
/* C[M,N] = op(A).op(B); ta/tb: 0 = as stored, 1 = transposed; leading dimensions in floats */
int eagcn_gemm_f32(int ta, int tb, int M, int N, int K, const float* A, int lda, const float* B,
                   int ldb, float* C, int ldc, void* stream);
/* the same product with one extent read from device memory (M / K are then capacities): which = 0 the rows of A and C,
   which = 2 the reduction length -- products over the packed rows of a capacity-sized batch index (eagcn_batch.meta[0]) */
int eagcn_gemm_f32_dev(int ta, int tb, int M, int N, int K, const float* A, int lda, const float* B,
                       int ldb, float* C, int ldc, const int32_t* extent_dev, int which, void* stream);

/* ---- persistent, balanced ("stream-K") form of the same products: the kernel the layer products run on ---------
 * A launch is a fixed grid; the iteration space (tiles x k-tiles, counted on the device) is cut into equal ranges, tiles
 * cut between workgroups are finished in the launch (no split-K slabs, no reduction launch).  Needs 16-byte aligned
 * operands, leading dimensions / contiguous extents that are multiples of 4, and eagcn_gemm_sk_workspace_bytes() bytes
 * of device scratch (need not be initialised).  (ta,tb) in {(0,0), (0,1), (1,0)}. */
size_t eagcn_gemm_sk_workspace_bytes(void);
int eagcn_gemm_f32_sk(int ta, int tb, int M, int N, int K, const float* A, int lda, const float* B, int ldb, float* C,
                      int ldc, void* workspace, size_t workspace_bytes, void* stream);
/* C0[M0,N0] = A0[M0,K0].B0[N0,K0]^T and C1[M1,N1] = A1[K1,M1]^T.B1[K1,N1] in ONE launch: the two autograd products of
 * reference layers.py:40 (dX = dP.W^T, dW = X^T.dP) */
int eagcn_gemm_pair_sk(int M0, int N0, int K0, const float* A0, int lda0, const float* B0, int ldb0, float* C0, int ldc0,
                       int M1, int N1, int K1, const float* A1, int lda1, const float* B1, int ldb1, float* C1, int ldc1,
                       void* workspace, size_t workspace_bytes, void* stream);
/* The same pair on the XCD-local schedule used by the layer backward: the long reduction of the second product is split across
 * the eight XCDs (each XCD reads only its own eighth of the K rows, for both products); C1 receives 8 partial slabs, `slab`
 * floats apart, whose sum (in slab order) is the product. */
int eagcn_gemm_pair_sk_slabs(int M0, int N0, int K0, const float* A0, int lda0, const float* B0, int ldb0, float* C0, int ldc0,
                             int M1, int N1, int K1, const float* A1, int lda1, const float* B1, int ldb1, float* C1, int ldc1,
                             size_t slab, void* workspace, size_t workspace_bytes, void* stream);
/* host-side mirror of that schedule's partition: a[9] = row-block (64 rows) boundaries of the first product per XCD segment,
 * c[9] = k-step (16 rows) boundaries of the second; wgs = 0: the library's grid */
int eagcn_gemm_sk_plan(int M0, int N0, int K0, int M1, int N1, int K1, int wgs, int* a, int* c);
int eagcn_gemm_sk_timeouts(void);   /* hand-offs that gave up waiting since load (must stay 0); synchronising device read */
/* A hand-off that times out is FATAL, never silent: the owner wave poisons its output tile with NaN and stores 1 into a
 * sticky word in host-mapped memory.  eagcn_gemm_sk_failed() reads that word without synchronising (the graph-replay host
 * loop polls it every step, eagcn_amd/graph.py); every eagcn_model_* / eagcn_layer_* entry point returns EAGCN_ERR_HIP while
 * it is set.  eagcn_gemm_sk_reset_failed() clears it; eagcn_gemm_sk_inject_failure() sets it (test hook). */
int eagcn_gemm_sk_failed(void);
void eagcn_gemm_sk_reset_failed(void);
void eagcn_gemm_sk_inject_failure(void);

/* ---- optional per-kernel-class timing with HIP events on the launch stream (bench.py roofline) -- */
void eagcn_prof_enable(int on);
void eagcn_prof_reset(void);
int eagcn_prof_ntags(void);
const char* eagcn_prof_tag_name(int tag);
/* sums event-pair durations of class `tag` since the last reset (waits for them); work = summed
 * algorithmic flops for the gemm class */
int eagcn_prof_read(int tag, double* total_ms, double* work, int64_t* launches);

#ifdef __cplusplus
}
#endif
#endif /* EAGCN_HIP_H */
