/* ptv3_hip.h -- C ABI of libptv3_hip.so: the MI355X (gfx950) kernels behind the PTv3
 * serialized-window attention path of Pointcept-KeypointDetection.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless the name ends in _host;
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream); every call only enqueues
 *     work on that stream: no allocation, no synchronisation, graph-capturable;
 *   - `dtype`: PTV3_F32 (0) = fp32 storage, exact-fp32 matrix-core math (parity mode);
 *              PTV3_BF16 (1) = bf16 storage, bf16 MFMA with fp32 accumulate / fp32 softmax+norm;
 *   - return value: 0 on success, non-zero on error (ptv3_last_error() gives the message);
 *   - tensors are dense row-major.  "rows" of feature matrices are points.
 *
 * Each entry point names the reference interface it replaces (paths relative to the reference
 * repository root).  The Python binding a maintainer adds is in INTEGRATION.md.
 */
#ifndef PTV3_HIP_H
#define PTV3_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PTV3_F32 0
#define PTV3_BF16 1

/* curve ids for ptv3_sfc_encode (pointcept/models/utils/serialization/default.py:9-24) */
#define PTV3_ORDER_Z 0
#define PTV3_ORDER_Z_TRANS 1
#define PTV3_ORDER_HILBERT 2
#define PTV3_ORDER_HILBERT_TRANS 3

/* epilogue activation ids for ptv3_gemm */
#define PTV3_ACT_NONE 0
#define PTV3_ACT_GELU 1 /* erf form = torch.nn.GELU() default */
#define PTV3_ACT_RELU 2

const char* ptv3_last_error(void);
int ptv3_version(void);

/* ---- serialization ------------------------------------------------------------------------
 * replaces encode() (utils/serialization/default.py:9-24), z_order.xyz2key (z_order.py:66-101)
 * and hilbert.encode (hilbert.py:91-198): code[r][i] = batch[i] << 3*depth | curve_r(grid_coord[i]).
 * grid_coord: (n,3) int32 (coord_is_i64=0) or int64 (=1); batch: (n) int64 or NULL;
 * order_ids_host: k curve ids (host array); code: (k,n) int64. */
int ptv3_sfc_encode(const void* grid_coord, int coord_is_i64, const int64_t* batch, int64_t n,
                    int depth, const int* order_ids_host, int k, int64_t* code, void* stream);

/* replaces torch.argsort + scatter_ of arange in Point.serialization
 * (models/utils/structure.py:92-99) and SerializedPooling (point_transformer_v3m1_base.py:399-406):
 * per row r: order[r] = stable argsort(code[r]); inverse[r][order[r][i]] = i.
 * code: (k,n) int64 compared as UNSIGNED 64-bit with all set bits below `end_bit` (<= 64; serialization codes
 * are non-negative, the GridSample voxel hashes use all 64 bits); order/inverse: (k,n) int64.
 * workspace: ptv3_argsort_workspace_bytes(k,n) bytes. */
size_t ptv3_argsort_workspace_bytes(int k, int64_t n);
int ptv3_argsort_i64(const int64_t* code, int k, int64_t n, int end_bit, int64_t* order,
                     int64_t* inverse, void* workspace, size_t workspace_bytes, void* stream);

/* ---- patch partition ----------------------------------------------------------------------
 * replaces SerializedAttention.get_padding_and_inverse (point_transformer_v3m1_base.py:114-170).
 * offset: (b) int64 cumulative scene ends (device); patch = K.  A scene of n_b > K points is padded to a
 * multiple of K (pad-by-borrowing, :144-154) and cut into K-slot windows; a scene of n_b <= K points stays
 * ONE window of n_b slots (:131-133 - only reachable with enable_flash=True, where K is fixed; with
 * enable_flash=False K = min(smallest scene, patch) and every window has K slots).
 * n_pad = sum_b (n_b > K ? ceil_to_K(n_b) : n_b);  windows = sum_b ceil(n_b / K).
 * pad: (n_pad) int64, unpad: (n) int64, cu_seqlens: (windows + 1) int32 window starts + n_pad. */
int ptv3_pad_plan(const int64_t* offset, int b, int64_t n, int64_t n_pad, int patch, int64_t* pad,
                  int64_t* unpad, int32_t* cu_seqlens, void* stream);

/* composes the two gathers of SerializedAttention.forward (point_transformer_v3m1_base.py:184-185):
 * win_order[p] = order[pad[p]] (n_pad), win_inverse[i] = unpad[inverse[i]] (n); int32 outputs. */
int ptv3_window_maps(const int64_t* order, const int64_t* inverse, const int64_t* pad,
                     const int64_t* unpad, int64_t n, int64_t n_pad, int32_t* win_order,
                     int32_t* win_inverse, void* stream);

/* pad plan + both maps for all k orders in one launch (no pad / unpad materialised): the executor's form of
 * the two entry points above.  order / inverse: (k, n); win_order: (k, n_pad); win_inverse: (k, n);
 * cu_seqlens (optional, NULL = skip): (windows + 1) int32 as ptv3_pad_plan writes it. */
int ptv3_window_plan(const int64_t* order, const int64_t* inverse, const int64_t* offset, int b, int k,
                     int64_t n, int64_t n_pad, int patch, int32_t* win_order, int32_t* win_inverse,
                     int32_t* cu_seqlens, void* stream);

/* ---- window attention ----------------------------------------------------------------------
 * replaces the vanilla branch of SerializedAttention.forward (point_transformer_v3m1_base.py:188-216)
 * = what flash_attn_varlen_qkvpacked_func computes on the reference's CUDA path (:208-214):
 *   x = qkv[win_order]; per window of `patch` rows and head h:
 *   out_w = softmax(scale * q k^T) v;  out[i] = concat_h(out_w)[win_inverse[i]].
 * qkv: (n, 3*c) with channel layout [3][heads][c/heads]; out: (n, c).  c/heads must be 16, 32 or 64.
 * rpe_bias (optional, may be NULL): (n_pad/patch, heads, patch, patch) fp32 added to the scores
 * (RPE.forward, :29-48). */
int ptv3_window_attn_fwd(const void* qkv, const int32_t* win_order, const int32_t* win_inverse,
                         void* out, int64_t n, int64_t n_pad, int c, int heads, int patch, float scale,
                         const float* rpe_bias, int dtype, void* stream);

/* The enable_flash=True form of the same attention: what flash_attn_varlen_qkvpacked_func(qkv[order], cu_seqlens,
 * max_seqlen=K, softmax_scale) computes at the reference's call site (point_transformer_v3m1_base.py:207-215;
 * flash-attn 2.6.3 is a CUDA-only wheel outside the reference tree - its published semantics are restated:
 * independent softmax(scale q k^T) v per sequence [cu_seqlens[w], cu_seqlens[w+1]) and head).  Windows may be
 * ragged: window w holds the padded slots [cu_seqlens[w], cu_seqlens[w+1]), 1..max_seqlen of them (a scene with
 * fewer than K points is one short window, :131-133).  cu_seqlens: (num_windows + 1) int32, device, as
 * ptv3_pad_plan / ptv3_window_plan write it.  sum_len_sq (host, measurement only): sum over windows of len^2
 * for the algorithmic flop count, 0 = unknown (n_pad * max_seqlen is used).  Arithmetic follows `dtype` like
 * ptv3_window_attn_fwd (the reference casts qkv to bf16 for flash-attn whatever the model dtype, :209). */
int ptv3_window_attn_varlen_fwd(const void* qkv, const int32_t* win_order, const int32_t* win_inverse,
                                const int32_t* cu_seqlens, int num_windows, void* out, int64_t n, int64_t n_pad,
                                int c, int heads, int max_seqlen, float scale, double sum_len_sq, int dtype,
                                void* stream);

/* Same attention with the relative-position bias of RPE (point_transformer_v3m1_base.py:29-48 applied to
 * get_rel_pos :104-112) evaluated INSIDE the kernel: bias(q, k, h) = sum over the 3 axes of
 * rpe_table[axis * (2*pos_bnd+1) + clamp(grid[q][axis] - grid[k][axis], -pos_bnd, pos_bnd) + pos_bnd][h],
 * from the window's voxel coordinates and the head's table column held in LDS - the (windows, H, K, K) bias
 * tensor of the rpe_bias argument above (3.3 GB per call at 100k points, K = 1024) is never built.
 * grid_coord (n, 3) int32 in point order; rpe_table (3*(2*pos_bnd+1), heads) fp32.  PTV3_ERR_UNSUPPORTED when the
 * window does not fit the resident-window kernel (then build the dense bias and use ptv3_window_attn_fwd). */
int ptv3_window_attn_rpe_fwd(const void* qkv, const int32_t* win_order, const int32_t* win_inverse, void* out,
                             int64_t n, int64_t n_pad, int c, int heads, int patch, float scale,
                             const int32_t* grid_coord, const float* rpe_table, int pos_bnd, int dtype,
                             void* stream);

/* ---- sparse submanifold convolution + dense linear (one implicit-GEMM kernel) ---------------
 * Hash of the active sites: replaces spconv's indice-pair generation behind
 * spconv.SparseConvTensor / SubMConv3d(indice_key=...) (models/utils/structure.py:111-146,
 * point_transformer_v3m1_base.py:277-284,499-506).
 * indices: (n,4) int32 [batch,x,y,z] unique rows, 0 <= x,y,z < 65536, batch < 32768.
 * table: `slots` int64 keys + `slots` int32 values, slots = power of two >= 2n
 * (ptv3_subm_table_slots(n)); layout: keys first.  nbr: (n, kvol) int32, -1 = inactive;
 * kernel offset index (a,b,c) -> a*k*k + b*k + c pairs with (x+a-k/2, y+b-k/2, z+c-k/2). */
int64_t ptv3_subm_table_slots(int64_t n);
int ptv3_subm_build_table(const int32_t* indices, int64_t n, void* table, int64_t slots, void* stream);
int ptv3_subm_neighbors(const int32_t* indices, int64_t n, const void* table, int64_t slots, int ksize,
                        int32_t* nbr, void* stream);

/* y = epilogue( sum_{d<kvol} sum_{c<cin} w[o][d][c] * x[nbr[i][d]][c] ).
 *   nbr == NULL, kvol == 1  ->  plain torch.nn.Linear: y = x @ w^T            (w: (cout, cin))
 *   nbr != NULL             ->  spconv SubMConv3d, w: (cout, k,k,k, cin) = (cout, kvol, cin)
 * epilogue, in this order, every pointer optional (NULL = skip):
 *   + bias[o];  * bn_scale[o] + bn_shift[o]  (eval BatchNorm1d folded);  act;
 *   y2 = y + res[res_index ? res_index[i] : i][o]   (written to `out2` if given, else into `out`);
 * `out` always receives the pre-residual value when out2 != NULL.
 * row_order (optional, (m) int32): output tile t processes rows row_order[64t..] (locality only).
 * x: (rows_x, cin) dtype - with nbr != NULL, rows_x = m: x is the feature matrix of the m active sites the table was
 * built for (ptv3_subm_neighbors), which lets the kernels address it through a 32-bit buffer descriptor;
 * out/out2/res: (m|rows, cout) dtype; w: dtype; bias/bn_*: fp32.
 * cin % 4 == 0 (fp32) / cin % 8 == 0 (bf16): 16-byte K granularity.
 * workspace (optional): ptv3_gemm_workspace_bytes(...) bytes of split-K slab room; shapes with few
 * rows and a long K (deep-stage convolutions) are then split over K and summed in slab order. */
size_t ptv3_gemm_workspace_bytes(int64_t m, int cin, int cout, int kvol, int dtype);
/* number of K slabs the shape is split into (1 = single pass).  With out == NULL ptv3_gemm leaves the raw fp32
 * slabs [splits][m][cout] in `workspace` for a fused consumer (ptv3_block_head). */
int ptv3_gemm_splits(int64_t m, int cin, int cout, int kvol, int dtype);
int ptv3_gemm(const void* x, const void* w, void* out, int64_t m, int cin, int cout, int kvol,
              const int32_t* nbr, const int32_t* row_order, const float* bias, const float* bn_scale,
              const float* bn_shift, int act, const void* res, const int32_t* res_index, void* out2,
              int dtype, void* workspace, size_t workspace_bytes, void* stream);

/* ---- fused row-local halves of Block.forward (point_transformer_v3m1_base.py:318-338) ----------------
 * ptv3_block_fusable(c, hidden, dtype, m) names the variant the two entry points below will run:
 *   1  c in {32,64} (the large-M levels): one wave carries 16 points through the whole chain in registers
 *      (no LDS, no barrier). bf16: wqkv, w1 (input dim c) and w2 (input dim hidden) must have their input
 *      channels permuted inside every 32-chunk as [0-3,16-19,4-7,20-23,8-11,24-27,12-15,28-31]
 *      (see csrc/block_fused.hip); fp32: natural order.
 *   2  c in {128,256,512}, hidden == 4c: one workgroup per 16 points, its waves split the output channels of
 *      every GEMM, activations hop through LDS; natural weight layout. Every workgroup streams all 12c^2
 *      weights, so the variant is only chosen up to a per-c row limit (default 16384 rows at c=128, off at
 *      256/512; env PTV3_COOP_ROWS_<c>); m = 0 asks for the capability alone.
 *   0  neither (use ptv3_gemm / ptv3_layernorm).
 *   head: x = conv output (or sum of its split-K slabs + conv_bias);
 *         f1 = LN(x; g0,b0) + shortcut;  qkv = LN(f1; g1,b1) @ wqkv^T + bqkv               (:319-324, :188)
 *   tail: f2 = attn @ wproj^T + bproj + f1;  out = f2 + fc2(GELU(fc1(LN(f2; g2,b2))))       (:219, :326-334) */
int ptv3_block_fusable(int c, int hidden, int dtype, int64_t m);
/* One linear layer on whole rows with its LayerNorm neighbours folded in (csrc/block_wide.hip, rows_linear_kernel):
 *   out = act(prologue(x) @ w^T + bias) [+ res],   x (m, c), w (cout, c) natural layout, out / res (m, cout)
 *   g0 == NULL, g1 == NULL : prologue = identity                                   (attn.proj + shortcut, :219)
 *   g0 == NULL, g1 != NULL : prologue = LayerNorm(x; g1, b1)                       (norm2 -> fc1 -> GELU, :330-333)
 *   g0 != NULL             : f1 = LayerNorm(x; g0, b0) + shortcut, stored to `f1`, then LayerNorm(f1; g1, b1)
 *                            (cpe LayerNorm + shortcut -> norm1 -> qkv, :319-324, :188)
 * Every row is read once, the weights stream through LDS (LDS-DMA ring); replaces ptv3_layernorm + ptv3_gemm for
 * c in {128, 256, 512} (fp32: up to 256), cout a multiple of 64 | 32.  ptv3_rows_linear_capable tells whether a shape
 * is served; WHEN to prefer it over the tiled GEMM (row count) is the caller's policy. */
int ptv3_rows_linear_capable(int c, int cout, int dtype, int64_t m);
int ptv3_rows_linear(const void* x, const void* shortcut, const float* g0, const float* b0, const float* g1,
                     const float* b1, const void* w, const float* bias, int act, const void* res, void* f1, void* out,
                     int64_t m, int c, int cout, float eps, int dtype, void* stream);
int ptv3_block_head(const void* x, const float* slab, int splits, const float* conv_bias, const void* shortcut,
                    const float* g0, const float* b0, const float* g1, const float* b1, const void* wqkv,
                    const float* bqkv, void* f1, void* qkv, int64_t m, int c, float eps, int dtype, void* stream);
/* Row-local two-layer MLP, hidden layer in registers (the dense keypoint head, offset_keypoint_ptv3.py:26-31:
 * Linear -> BatchNorm1d(eval) -> ReLU -> Linear):  out = act((x @ w1^T + b1) * s1 + t1) @ w2^T + b2.
 * x (m, cin) dtype, cin in {32, 64}; w1 (hidden, cin) natural; b1 / s1 / t1 (hidden) fp32 (each optional: 0 / 1 / 0);
 * w2 (16 | 32 | 64 rows for cout <= 16 | 32 | 64, hidden) with zero rows beyond cout and, for bf16, its input channels chain-permuted exactly like
 * ptv3_block_tail's w2; b2 (cout) fp32; out (m, cout) fp32 when out_f32 else dtype.  hidden % 64 == 0, cout % 4 == 0,
 * cout <= 64 (ptv3_mlp2_fusable). */
int ptv3_mlp2_fusable(int cin, int hidden, int cout, int dtype);
int ptv3_mlp2(const void* x, const void* w1, const float* b1, const float* s1, const float* t1, int act,
              const void* w2, const float* b2, void* out, int out_f32, int64_t m, int cin, int hidden, int cout,
              int dtype, void* stream);
int ptv3_block_tail(const void* attn, const void* f1, const void* wproj, const float* bproj, const float* g2,
                    const float* b2, const void* w1, const float* bias1, const void* w2, const float* bias2,
                    void* out, int64_t m, int c, int hidden, float eps, int dtype, void* stream);

/* ---- normalisation / elementwise -------------------------------------------------------------
 * torch.nn.LayerNorm over the last dim (Block.cpe[2], norm1, norm2; :277-304): y = LN(x)*g+b [+ res];
 * optional second output y2 = LN2(y) * g2 + b2 (fuses `shortcut + cpe` with the following norm1). */
int ptv3_layernorm(const void* x, const float* gamma, const float* beta, const void* res, void* y,
                   const float* gamma2, const float* beta2, void* y2, int64_t m, int c, float eps,
                   int dtype, void* stream);

/* same, with x given as the split-K fp32 slabs a ptv3_gemm(out = NULL) left behind: x = T(sum_z slab[z] + slab_bias)
 * (saves the separate reduce launch and one (m, c) round trip in front of the xCPE LayerNorm, :283) */
int ptv3_layernorm_slabs(const float* slab, int splits, const float* slab_bias, const float* gamma, const float* beta,
                         const void* res, void* y, const float* gamma2, const float* beta2, void* y2, int64_t m, int c,
                         float eps, int dtype, void* stream);

/* y = act(x * scale[c] + shift[c]) : eval BatchNorm1d + GELU of Embedding / SerializedPooling
 * (:508-511, 439-442). In-place allowed. */
int ptv3_affine_act(const void* x, const float* scale, const float* shift, int act, void* y, int64_t m,
                    int c, int dtype, void* stream);

/* dtype conversion helpers (fp32 <-> bf16 storage) */
int ptv3_cast(const void* x, int src_dtype, void* y, int dst_dtype, int64_t count, void* stream);

/* ---- serialized pooling ("grid-pool scatter") ------------------------------------------------
 * replaces torch.unique / torch.sort / cumsum / segment_csr in SerializedPooling.forward
 * (point_transformer_v3m1_base.py:384-428).  Points are visited in serialized order 0, so clusters
 * (equal code0 >> 3*pooling_depth) are contiguous runs of order0.
 * step 1 (segments): cluster[i] (n) int64 = rank of the parent code; seg_start (n+1) int32 run starts
 *         in order0 positions (first *n_out+1 entries valid); n_out written to device AND the
 *         caller reads it back (the one host sync per pooling, as torch.unique has).
 *         batch (n) int64 + pooled_offset (b) int64 (both optional): cumulative scene ends of the
 *         pooled Point (= batch2offset(batch[head]), models/utils/misc.py:32-34); pooled_offset[b-1]
 *         equals n_out.  workspace: ptv3_pool_workspace_bytes(n). */
size_t ptv3_pool_workspace_bytes(int64_t n);
int ptv3_pool_segments(const int64_t* code0, const int64_t* order0, int64_t n, int shift_bits,
                       const int64_t* batch, int64_t* cluster, int32_t* seg_start, int32_t* n_out,
                       int64_t* pooled_offset, void* workspace, size_t workspace_bytes, void* stream);
/* step 2 (reduce): for pooled row j over members order0[seg_start[j] .. seg_start[j+1]):
 *   feat_out[j]  = act(max_members(feat[.]) * bn_scale + bn_shift)        (n_out, c) dtype
 *   coord_out[j] = mean_members(coord[.])                                  (n_out, 3) fp32
 *   head = first member: grid_out[j] = grid_coord[head] >> pooling_depth (int64), batch_out[j] =
 *   batch[head], code_out[r][j] = code[perm[r]][head] >> 3*pooling_depth for r < k, where perm is the
 *   order shuffle of :408-412 (row_perm_host, k host ints; NULL = identity).
 *   feat == NULL skips the feature half, grid_coord == NULL skips the geometry half (the executor runs the
 *   two halves on different streams). */
int ptv3_pool_reduce(const void* feat, const float* coord, const int64_t* grid_coord,
                     const int64_t* batch, const int64_t* code, int k, const int64_t* order0,
                     const int32_t* seg_start, int64_t n, int64_t n_out, int c, int pooling_depth,
                     const float* bn_scale, const float* bn_shift, int act, const int* row_perm_host,
                     void* feat_out, float* coord_out, int64_t* grid_out, int64_t* batch_out,
                     int64_t* code_out, int dtype, void* stream);

/* ---- whole-model forward ------------------------------------------------------------------------
 * PointTransformerV3.forward (point_transformer_v3m1_base.py:699-714) in eval mode, optionally followed by
 * the dense keypoint-offset head (offset_keypoint_ptv3.py:26-31), as one call: every kernel above is
 * launched back to back from native code.  Used by pointcept.models (PT-v3m1) as its fast path; results
 * are identical to composing the per-op entry points.
 * params: flat table of device pointers in module order -
 *   stem conv w, stem BN scale, shift;
 *   per encoder stage s: [s>0: down.proj w, b, BN scale, shift] then per block the 16 pointers
 *     cpe conv w, b (with the cpe Linear folded in: W'_d = W_lin W_d, b' = W_lin b_conv + b_lin),
 *     cpe LN g, b, norm1 g, b, qkv w, b, proj w, b, norm2 g, b, fc1 w, b, fc2 w, b;
 *   per decoder stage s = S-2..0: up.proj w, b, BN scale, shift, up.proj_skip w, b, BN scale, shift, blocks;
 *   head (if head_out > 0): linear0 w, b, BN scale, shift, linear1 w, b.
 *   matrices in `dtype` (conv weights (cout, kvol, cin) with cin padded to the 16-byte granule),
 *   vectors fp32, BatchNorm folded to (scale, shift). */
typedef struct {
  int32_t dtype, in_channels /* padded */, num_stages, num_orders;
  int32_t enc_depths[8], enc_channels[8], enc_heads[8], enc_patch[8];
  int32_t dec_depths[8], dec_channels[8], dec_heads[8], dec_patch[8];
  int32_t stride[8];
  int32_t enc_mode, enable_flash; /* enable_flash = 0: patch = min(smallest scene, patch) (:173-176) */
  int32_t head_hidden, head_out;  /* head_out = 0: backbone only */
  float mlp_ratio, ln_eps, qk_scale /* 0: (c/heads)^-0.5 */;
} ptv3_model_desc;

typedef struct {
  const void* grid_coord; int32_t coord_is_i64; /* (n,3) */
  const void* feat;                             /* (n, in_channels) dtype */
  const int64_t* batch;                         /* (n), or NULL: derived from offset (misc.py:25-30) */
  const int64_t* offset;                        /* (b) cumulative, device */
  const int64_t* offset_host;                   /* (b) same values on the host, or NULL: read back internally */
  int32_t b; int64_t n;
  int32_t depth;                                /* bit_length(max(grid_coord)+1) (structure.py:73); 0: computed */
  const int32_t* order_ids_host;                /* num_orders curve ids, already in shuffled order */
  const int32_t* pool_perm_host;                /* (num_stages-1, num_orders) row permutations (:408-412) */
  int64_t *code, *order, *inverse;              /* out: (num_orders, n) level-0 serialization */
  void* out_feat;                               /* out: (n, dec_channels[0]) dtype */
  float* out_head;                              /* out: (n, head_out) fp32 (head_out > 0) */
  int64_t* stage_points_host;                   /* out, optional: num_stages point counts */
  int32_t* depth_out;                           /* out, optional: the serialization depth used */
  int64_t* batch_out;                           /* out, optional (batch == NULL): (n) derived batch ids */
  int32_t inputs_resident;                      /* 1: grid_coord / batch / offset were complete in memory before
                                                   this call (not produced on `stream` just now): the geometry
                                                   pipeline may then overlap the previous call's feature tail */
  int32_t overlap_calls;                        /* 1 (needs inputs_resident, parameters unchanged since an earlier
                                                   synchronised call): the feature pipeline runs on one of two
                                                   executor-owned streams instead of `stream`, so the small, latency-
                                                   bound deep levels of call i execute under the chip-filling level-0
                                                   kernels of call i+1.  `stream` only receives a wait on this call's
                                                   completion.  Contract: out_feat / out_head must not be buffers whose
                                                   last readers were enqueued on `stream` after the PREVIOUS call was
                                                   issued (use a ring of >= 3 output buffers, consume call i's outputs
                                                   before issuing call i+2). */
  const void* raw_feat;                         /* overlap mode, optional: (n, raw_feat_channels) features still in   */
  int32_t raw_feat_channels;                    /* their original dtype / width; the executor pads them to            */
  int32_t raw_feat_dtype;                       /* desc->in_channels and casts to desc->dtype on its own stream       */
  int64_t arena_n;                              /* > 0: the workspace was sized by ptv3_forward_workspace_bytes(desc,  */
  int32_t arena_b;                              /* arena_n, arena_b) with arena_n >= n, arena_b >= b: the four arena    */
                                                /* parts then sit at offsets that depend on (arena_n, arena_b) only.   */
                                                /* REQUIRED whenever calls may still be in flight (inputs_resident /    */
                                                /* overlap_calls) and scene sizes vary: with 0 the parts are laid out   */
                                                /* for THIS call's n and would move under the previous call's kernels.  */
  void* executor;                               /* ptv3_executor_create() handle, or NULL = the current device's default */
                                                /* executor.  An executor owns the internal streams / events / call     */
                                                /* parity; calls on one executor are ordered as described above, calls  */
                                                /* on different executors (other models, other devices) are independent. */
} ptv3_forward_io;

/* Executor handles (optional).  create() binds to the CURRENT device; destroy() drains the executor's streams.
 * Every workspace handed to ptv3_forward with inputs_resident / overlap_calls must stay with ONE executor. */
void* ptv3_executor_create(void);
int ptv3_executor_destroy(void* executor);

size_t ptv3_forward_workspace_bytes(const ptv3_model_desc* desc, int64_t n, int b);  /* sized for overlap_calls too */
int ptv3_forward(const ptv3_model_desc* desc, const void* const* params, int num_params,
                 const ptv3_forward_io* io, void* workspace, size_t workspace_bytes, void* stream);

/* ---- training: backward kernels ---------------------------------------------------
 * The reference has no hand-written backward on this path: torch autograd differentiates the forward
 * statements cited at each forward entry point above.  These entry points are what a torch.autograd.Function
 * per layer calls (pointcept-keypointdetection_amd/ptv3_hip/autograd.py).  All reductions over points use
 * fixed row chunks -> fp32 slabs -> ordered sums (deterministic, no atomics).
 *
 * dw (cout, kvol*cin) fp32 = dy^T (m, cout) . gather(x (m, cin) through nbr (m, kvol))   [kvol=1: nbr NULL]
 *   = weight gradient of nn.Linear (point_transformer_v3m1_base.py:188,219,232-244) and of SubMConv3d
 *   (:277-284, 499-506) in the layout of ptv3_gemm's w.  Input gradients reuse ptv3_gemm itself: linear with
 *   w^T; sparse conv with w'[c][t][o] = w[o][kvol-1-t][c] (submanifold neighbour maps are symmetric).
 * dbias (cout) fp32 or NULL: the bias gradient sum_i dy[i][:] from the same pass (the staging of dy sees every
 *   element once; a separate column reduction would read dy again and cost two more launches per layer). */
size_t ptv3_gemm_tn_workspace_bytes(int64_t m, int cout, int cin, int kvol);
int ptv3_gemm_tn(const void* dy, const void* x, const int32_t* nbr, float* dw, float* dbias, int64_t m, int cout,
                 int cin, int kvol, int dtype, void* workspace, size_t workspace_bytes, void* stream);
/* column reductions over the m rows of a (and b), out fp32:
 *   mode 0: out[c]       = sum a                          (bias gradients)
 *   mode 1: out[2][c]    = sum a, sum a*a                 (BatchNorm1d batch statistics, :439-441, 508-510)
 *   mode 2: out[2][c]    = sum a, sum a*(b - mu)*rs       (BatchNorm1d backward: a = dy, b = x)
 *   mode 3: out[2][c]    = sum a, sum (a - mu)^2          (centred second pass of the batch variance) */
size_t ptv3_col_reduce_workspace_bytes(int64_t m, int c);
int ptv3_col_reduce(const void* a, const void* b, const float* mu, const float* rs, float mu_scale, int mode, float* out,
                    int64_t m, int c, int dtype, void* workspace, size_t workspace_bytes, void* stream);
/* modes 2 and 3 subtract mu[c] * mu_scale: with mu = the column sums of a first pass and mu_scale = 1 / m the mean never
 * exists as a tensor.
 * The per-channel arithmetic of a BatchNorm1d training step (:439-441, 508-510, the head of offset_keypoint_ptv3.py) in one
 * launch each.  ptv3_bn_finalize: sum / centred_sq (c) = the two col_reduce passes; writes mean, rstd, scale = weight rstd,
 * shift = bias - mean scale and updates the running buffers (NULL: none) with `momentum` and the unbiased variance, as
 * torch.nn.BatchNorm1d does.  ptv3_bn_bwd_coeffs: sums (2, c) = col_reduce mode 2 of the pre-activation gradient; writes
 * the coefficients of ptv3_affine2 (dx = ca dy + cb x + cc). */
int ptv3_bn_finalize(const float* sum, const float* centred_sq, int64_t m, const float* weight, const float* bias,
                     float* running_mean, float* running_var, float momentum, float eps, float* mean, float* rstd,
                     float* scale, float* shift, int c, void* stream);
int ptv3_bn_bwd_coeffs(const float* sums, int64_t m, const float* weight, const float* rstd, const float* mean, float* ca,
                       float* cb, float* cc, int c, void* stream);
/* LayerNorm backward (statistics recomputed from x): dx (m, c) dtype, dgamma_dbeta (2, c) fp32.
 * add (m, c) dtype or NULL: dx = add + (input gradient) - the gradient arriving over the residual connection around the
 * normalised branch (Block.forward, :318-338), folded into the store.  workspace: ptv3_col_reduce_workspace_bytes(m, c). */
int ptv3_layernorm_bwd(const void* x, const void* dy, const void* add, const float* gamma, float eps, void* dx,
                       float* dgamma_dbeta, int64_t m, int c, int dtype, void* workspace,
                       size_t workspace_bytes, void* stream);
/* dx = dy * act'(x * scale[c] + shift[c])  (scale = shift = NULL: act'(x)); act = PTV3_ACT_* */
int ptv3_act_bwd(const void* dy, const void* x, const float* scale, const float* shift, int act, void* dx,
                 int64_t m, int c, int dtype, void* stream);
/* dx = ca[c]*dy + cb[c]*x + cc[c]: BatchNorm1d input gradient with the batch statistics folded into ca/cb/cc */
int ptv3_affine2(const void* dy, const void* x, const float* ca, const float* cb, const float* cc, void* dx,
                 int64_t m, int c, int dtype, void* stream);
/* SerializedPooling segment-max backward (:416-421): the first member holding the maximum gets dy, others 0;
 * SerializedUnpooling gather backward (:480): out[j] = sum of dy over the members of segment j. */
int ptv3_pool_max_bwd(const void* feat, const void* dy, const int64_t* order0, const int32_t* seg_start,
                      int64_t n_out, int c, void* dfeat, int dtype, void* stream);
int ptv3_segment_sum(const void* dy, const int64_t* order0, const int32_t* seg_start, int64_t n_out, int c,
                     void* out, int dtype, void* stream);
/* window attention backward (:196-204 differentiated): qkv (n, 3c), out = forward output (n, c), dout (n, c)
 * -> dqkv (n, 3c).  Softmax statistics are recomputed (nothing saved by the forward). rpe bias unsupported. */
size_t ptv3_window_attn_bwd_workspace_bytes(int64_t n, int64_t n_pad, int c, int heads, int dtype);
int ptv3_window_attn_bwd(const void* qkv, const void* out, const void* dout, const int32_t* win_order,
                         const int32_t* win_inverse, void* dqkv, int64_t n, int64_t n_pad, int c, int heads,
                         int patch, float scale, int dtype, void* workspace, size_t workspace_bytes,
                         void* stream);
/* backward of ptv3_window_attn_varlen_fwd (ragged windows; same workspace size) */
int ptv3_window_attn_varlen_bwd(const void* qkv, const void* out, const void* dout, const int32_t* win_order,
                                const int32_t* win_inverse, const int32_t* cu_seqlens, int num_windows, void* dqkv,
                                int64_t n, int64_t n_pad, int c, int heads, int max_seqlen, float scale, int dtype,
                                void* workspace, size_t workspace_bytes, void* stream);
/* Training pair of the attention: the forward also leaves the log2-domain log-sum-exp of every (padded slot, head) row
 * (lse: n_pad x heads floats), the backward takes it instead of recomputing it in its first pass (a third of that pass's
 * work).  cu_seqlens NULL: uniform windows of `patch` slots, else ragged ones (patch = max_seqlen, sum_len_sq for the flop
 * count).  Same kernels and results as ptv3_window_attn_fwd / _varlen_fwd; dqkv differs from ptv3_window_attn_bwd's only
 * by the summation the two log-sum-exps went through.  workspace: ptv3_window_attn_bwd_workspace_bytes. */
int ptv3_window_attn_train_fwd(const void* qkv, const int32_t* win_order, const int32_t* win_inverse,
                               const int32_t* cu_seqlens, int num_windows, void* out, float* lse, int64_t n,
                               int64_t n_pad, int c, int heads, int patch, float scale, double sum_len_sq, int dtype,
                               void* stream);
int ptv3_window_attn_train_bwd(const void* qkv, const void* out, const void* dout, const float* lse,
                               const int32_t* win_order, const int32_t* win_inverse, const int32_t* cu_seqlens,
                               int num_windows, void* dqkv, int64_t n, int64_t n_pad, int c, int heads, int patch,
                               float scale, int dtype, void* workspace, size_t workspace_bytes, void* stream);
/* Attention dropout in training (reference: nn.Dropout on the attention probabilities,
 * point_transformer_v3m1_base.py:203; flash_attn dropout_p, :211): out_i = sum_j keep_ij softmax(s)_ij v_j / (1 - p).
 * keep_ij is a counter-based hash of (padded query slot, head, key slot in the window) and `seed` (csrc/common.h
 * drop_keep), regenerated by the backward from the same seed - nothing is stored.  The reference's Philox stream is
 * not reproduced: same distribution, different values.  cu_seqlens NULL: uniform windows of `patch` slots, else the
 * ragged layout of ptv3_window_attn_varlen_fwd (patch = max_seqlen).  0 < p_drop < 1.  No rpe bias.
 * The backward takes the forward's out, p_drop and seed; workspace as ptv3_window_attn_bwd_workspace_bytes. */
int ptv3_window_attn_drop_fwd(const void* qkv, const int32_t* win_order, const int32_t* win_inverse,
                              const int32_t* cu_seqlens, int num_windows, void* out, int64_t n, int64_t n_pad, int c,
                              int heads, int patch, float scale, float p_drop, uint32_t seed, int dtype, void* stream);
int ptv3_window_attn_drop_bwd(const void* qkv, const void* out, const void* dout, const int32_t* win_order,
                              const int32_t* win_inverse, const int32_t* cu_seqlens, int num_windows, void* dqkv,
                              int64_t n, int64_t n_pad, int c, int heads, int patch, float scale, float p_drop,
                              uint32_t seed, int dtype, void* workspace, size_t workspace_bytes, void* stream);
/* backward of ptv3_window_attn_rpe_fwd: dqkv as above with the relative-position bias inside the recomputed softmax,
 * plus the gradient of the bias table, dtable (3*(2*pos_bnd+1), heads) fp32 = sum over all (window, query, key) pairs
 * of dS routed to the three (axis, clamped coordinate difference) entries the pair read (RPE.forward,
 * point_transformer_v3m1_base.py:29-48, differentiated).  Per-workgroup partial columns are summed in a fixed order.
 * PTV3_ERR_UNSUPPORTED when 3*(2*pos_bnd+1) > 1024. */
size_t ptv3_window_attn_rpe_bwd_workspace_bytes(int64_t n, int64_t n_pad, int c, int heads, int patch, int pos_bnd,
                                                int dtype);
int ptv3_window_attn_rpe_bwd(const void* qkv, const void* out, const void* dout, const int32_t* win_order,
                             const int32_t* win_inverse, const int32_t* grid_coord, const float* rpe_table,
                             int pos_bnd, void* dqkv, float* dtable, int64_t n, int64_t n_pad, int c, int heads,
                             int patch, float scale, int dtype, void* workspace, size_t workspace_bytes, void* stream);
/* ---- one Block, training, as two native calls -------------------------------------------------------------------
 * Block.forward (point_transformer_v3m1_base.py:318-338, pre-norm, LayerNorm / GELU) and what torch autograd derives
 * from it, each as ONE C call issuing the same kernels as the per-op entry points above would (ptv3_hip/autograd.py
 * BlockFn composes those and is this descriptor's checker):
 *   c = LN0(lin(conv(conv_feat))); f1 = feat + c; f2 = f1 + mask1 * proj(attn(qkv(LN1(f1))));
 *   out = f2 + mask2 * fc2(GELU(fc1(LN2(f2))))
 * All (n, .) tensors in `dtype`, vectors / gradients of parameters fp32.  w_*: (cout, K) as ptv3_gemm's w; wt_*
 * (backward): W^T (cin, cout) of the linears, the mirrored-tap transposed weight (cin, kvol*cout) of the conv.
 * conv_feat NULL: the xCPE conv reads feat (every block but the first decoder block of a stage).  mask1 / mask2 (n) fp32
 * or NULL: uniform draws u; the per-point DropPath factor of a branch is u[i] < keep ? 1 / keep : 0 (timm drop_path with
 * scale_by_keep on the (N, C) matrix).  cu_seqlens NULL: uniform windows of `patch` slots.
 * fwd writes c1 .. out and attn_lse; bwd reads them plus dout and writes dfeat (and dconv_feat), dw_* (cout, K), db_* (cout),
 * dln0/1/2 (2, c) = [dgamma | dbeta] of the three LayerNorms.  workspace: ptv3_block_train_workspace_bytes(). */
typedef struct ptv3_block_train {
  int64_t n, n_pad;
  int32_t c, hidden, heads, patch, kvol, num_windows, dtype, reserved;
  float scale, eps, keep1, keep2;
  double sum_len_sq;
  const int32_t *nbr, *row_order, *win_order, *win_inverse, *cu_seqlens;
  const void *feat, *conv_feat;
  const void *w_conv, *w_lin, *w_qkv, *w_proj, *w_fc1, *w_fc2;
  const void *wt_conv, *wt_lin, *wt_qkv, *wt_proj, *wt_fc1, *wt_fc2;
  const float *b_conv, *b_lin, *b_qkv, *b_proj, *b_fc1, *b_fc2, *g0, *b0, *g1, *b1, *g2, *b2;
  const float *mask1, *mask2;
  void *c1, *c2, *f1, *t3, *qkv, *a, *f2, *t5, *h0, *h, *out;
  const void* dout;
  void *dfeat, *dconv_feat;
  float *dw_conv, *dw_lin, *dw_qkv, *dw_proj, *dw_fc1, *dw_fc2;
  float *db_conv, *db_lin, *db_qkv, *db_proj, *db_fc1, *db_fc2;
  float *dln0, *dln1, *dln2;
  void* workspace;
  size_t workspace_bytes;
  float* attn_lse;   /* (n_pad, heads) fp32: written by fwd, read by bwd (ptv3_window_attn_train_fwd / _bwd) */
} ptv3_block_train;
size_t ptv3_block_train_workspace_bytes(const ptv3_block_train* block, int backward);
int ptv3_block_train_fwd(const ptv3_block_train* block, void* stream);
int ptv3_block_train_bwd(const ptv3_block_train* block, void* stream);

/* fused multi-tensor AdamW (torch.optim.AdamW semantics; pointcept/utils/optimizer.py builds it with one
 * extra "block" parameter group): a device table of ptv3_adamw_entry_bytes()-sized entries, filled on the
 * host by ptv3_adamw_fill_entry; entry i owns blocks [first_block_i, first_block_i + ceil(numel_i / chunk)).
 * fp32 params / grads / moments.  grad_scale multiplies every gradient (1/loss-scale or a clip factor).
 * ptv3_grad_sqnorm: out[0] = sum of squares of all gradients (clip_grad_norm_), partial_ws (total_blocks). */
size_t ptv3_adamw_entry_bytes(void);
int ptv3_adamw_chunk(void);
int ptv3_adamw_fill_entry(void* entry_host, void* param, const void* grad, void* exp_avg, void* exp_avg_sq,
                          int64_t numel, int group, int first_block);
/* optional, after ptv3_adamw_fill_entry: the step also writes the updated parameter into `shadow` (same layout, fp32 or
 * bf16) and, when shadow_t is given, into a (cols, kvol mirrored, rows) transposed copy of the (rows, kvol, cols) weight
 * - W^T of a Linear (kvol 1) / the mirrored-tap weight of a SubMConv3d input gradient - so the training kernels never
 * cast or transpose weights between steps. */
int ptv3_adamw_fill_shadow(void* entry_host, void* shadow, void* shadow_t, int rows, int cols, int kvol,
                           int shadow_dtype);
/* optional, after ptv3_adamw_fill_entry: this tensor has taken `step_lag` steps fewer than the `step` argument of
 * ptv3_adamw_step (torch.optim.AdamW keeps state["step"] per parameter: a parameter that first receives a gradient
 * later, or a resumed checkpoint, engines/hooks/misc.py:269); its bias corrections use step - step_lag. */
int ptv3_adamw_fill_step_lag(void* entry_host, int64_t step_lag);
/* the transposed shadows are written by a second pass of 64 x 64 tiles: entry i owns tiles [first_tile_i, first_tile_i +
 * ptv3_adamw_shadow_tiles(rows, cols, kvol)) when it has a shadow_t and none otherwise (first_tile_i = the running sum,
 * set on EVERY entry); total_tiles / shadow_dtype of ptv3_adamw_step describe that pass (0 tiles: no shadow_t anywhere). */
int ptv3_adamw_shadow_tiles(int rows, int cols, int kvol);
int ptv3_adamw_fill_first_tile(void* entry_host, int first_tile);
/* grads_host (ntensors host array of device pointers) or NULL: this step's gradient of every table entry, overriding the
 * address stored in the table.  torch's autograd leaves a new gradient tensor on each parameter after
 * zero_grad(set_to_none=True) (no zero fill, no accumulate-add per parameter); the addresses travel as kernel arguments,
 * 256 tensors per launch, so the table is neither rebuilt nor re-uploaded.  first_block_host (ntensors): the
 * first_block given to ptv3_adamw_fill_entry for every entry (required with grads_host). */
int ptv3_adamw_step(const void* table_dev, int ntensors, int total_blocks, const float* lr_host,
                    const float* wd_host, int ngroups, float beta1, float beta2, float eps, int64_t step,
                    float grad_scale, const int32_t* first_block_host, const void* const* grads_host, int total_tiles,
                    int shadow_dtype, void* stream);
int ptv3_grad_sqnorm(const void* table_dev, int ntensors, int total_blocks, float* partial_ws, float* out,
                     const int32_t* first_block_host, const void* const* grads_host, void* stream);

/* ---- GridSample (the step before the model) -----------------------------------------------------------
 * Front half of pointcept/datasets/transform.py:848-860 for one cloud: grid_coord = floor(coord / grid_size)
 * evaluated in float64 as numpy does (float32 coord / float64 grid size), minus its per-axis minimum; key = the
 * reference's voxel hash of the shifted coordinate (:926-964).  coord (n,3) fp32; grid_coord (n,3) int64 out;
 * min_max (6) int64 out = [min xyz, max xyz] of the unshifted voxel coordinate; key (n) uint64 out.
 * The rest of GridSample composes existing entry points: ptv3_argsort_i64(key, end_bit 64) = np.argsort,
 * ptv3_pool_segments(key, order, shift 0) = np.unique(return_inverse, return_counts). */
#define PTV3_HASH_FNV 0
#define PTV3_HASH_RAVEL 1
int ptv3_grid_hash(const float* coord, int64_t n, double grid_size, int hash_type, int64_t* grid_coord,
                   int64_t* min_max, uint64_t* key, void* stream);

/* ---- keypoint aggregation (the step after the model) -------------------------------------------------
 * One launch for the per-sample x per-keypoint python loops of engines/hooks/offset_keypoint_evaluator.py:46-84
 * and tools/infer_offset.py:555-597.  coord (N,3) fp32; pred (N, nkp, 4) fp32 = [offset xyz, score] (model output
 * or target); offset (nscenes) cumulative int64; scale (nscenes) / centroid (nscenes,3) fp32 or NULL (1 / 0).
 * point(i) = coord[i]*scale + centroid + pred[i][k][0:3]*scale  (each step rounded as the torch statements).
 *   PTV3_KP_ARGMAX    kp = point(first arg max of the score)                    aux = that index within the scene
 *   PTV3_KP_WEIGHTED  kp = sum w point / sum w over score > thresh, arg max if none (infer_offset "weighted")
 *                                                                               aux = # points over thresh
 *   PTV3_KP_GT_MEAN   kp = mean of point(i) over score > 0 (evaluator :49-53)   aux = # such points, NaN if 0
 *   PTV3_KP_GT_FIRST  kp = point(first i with score > 0.5) (infer_offset :592-596)  aux = that index or -1 (NaN)
 * kp_out (nscenes, nkp, 3) fp32, aux_out (nscenes, nkp) int32. */
#define PTV3_KP_ARGMAX 0
#define PTV3_KP_WEIGHTED 1
#define PTV3_KP_GT_MEAN 2
#define PTV3_KP_GT_FIRST 3
int ptv3_keypoint_aggregate(const float* coord, const float* pred, const int64_t* offset, int nscenes, int nkp,
                            const float* scale, const float* centroid, int mode, float thresh, float* kp_out,
                            int32_t* aux_out, void* stream);

/* ---- measurement ---------------------------------------------------------------------------------
 * While enabled, the GEMM / fused-block / attention entry points (forward and backward) bracket their launches with
 * HIP events on the launch stream.  collect() synchronises the device and returns, per kernel family (0 linear,
 * 1 subm_conv, 2 window_attn, 3 backward: arrays of FOUR entries): summed device milliseconds, algorithmic flops
 * (sparse conv: 2*cin*cout per ACTIVE neighbour), algorithmic bytes and launch count since enable / the last collect. */
int ptv3_profile_enable(int on);
int ptv3_profile_collect(double* ms, double* flops, double* bytes, int64_t* launches);
/* Flops of the NEXT bracketed launch, for kernels whose work is decided by device data the entry point does not read
 * back (ptv3_swin_attn_fwd: the pair count sum_w len_w^2 lives in w_start); ignored while profiling is off. */
int ptv3_profile_hint_flops(double flops);
/* the same records by KERNEL (one launch per bracket): arrays of ptv3_profile_kernel_count() entries, entry i is the
 * kernel ptv3_profile_kernel_name(i) (gemm_kernel 64 / 32 channel tiles and gemm_big_kernel each split into their
 * dense and gathered = sparse-conv launches, the fused block halves, mlp2, the two window-attention kernels).  Does not
 * reset the records: call it BEFORE ptv3_profile_collect. */
int ptv3_profile_kernel_count(void);
const char* ptv3_profile_kernel_name(int kernel);
int ptv3_profile_collect_kernels(double* ms, double* flops, double* bytes, int64_t* launches);

/* ---- Swin3D window partition + cRSE window attention (row A19; PARITY UNPINNED) -----------------------------
 * The reference computes these through MinkowskiEngine and microsoft/Swin3D, neither of which is in its tree
 * (SURVEY.md section 8c); oracle/swin3d.py restates them from pointcept/models/swin3d/swin3d_layers.py and the
 * Swin3D paper and is the only checker these two entry points have.
 *
 * ptv3_swin_window_keys replaces the MinkowskiMaxPooling + coordinate_manager.kernel_map part of
 * BasicLayer.get_map_pair / get_window_mapping (swin3d_layers.py:715-795) and get_shifted_sp (:826-840):
 * coords (n,4) int32 [batch, x, y, z] at tensor stride `stride`; voxel = floor(c / stride) + shift; window =
 * floor(voxel / window_size) per axis; key = (batch, window xyz) << 9 | w_w_id with w_w_id = (lx*ws + ly)*ws + lz
 * (:781-788).  ptv3_argsort_i64(key, end_bit 64) is then the reference's sort by in_map (:752), and
 * ptv3_pool_segments(key, order, shift 9) yields the windows' token ranges (nempty_num, :777-778).
 * *bad is set non-zero when a batch index (0..4095) or window coordinate (-4096..4095) does not fit the key. */
int ptv3_swin_window_keys(const int32_t* coords, int64_t n, int stride, int window_size, int shift, int64_t* key,
                          int32_t* bad, void* stream);
/* replaces SelfAttnAIOFunction.apply(..., PosEmb.SEPARATE, TableDims.D0, IndexMode.INDIRECT, ...) at
 * swin3d_layers.py:556-569 (forward; ptv3_swin_attn_bwd below is its backward).  q, k, v, out: (n, heads, head_dim) in ORIGINAL voxel order, q already
 * scaled (:499); {q,k,v}_table: the concatenated fp32 tables of :503-528, table_offsets_host[c] elements per signal
 * axis c (the reference's `table_offsets`, :441/:452/:464), num_axes = 3, 6 or 9; n2n (n) int64: sorted position ->
 * original row (n2n_indices); w_start (num_windows + 1) int32: token range of each window in sorted order
 * (w2n_indices plus the total); n_crse (n, num_axes) fp32 in sorted order (:505-530); max_tokens >= the largest
 * window (window_size^3 always is).  head_dim 8, 16 or 32.
 *   e_ij = q_i.k_j + sum_c (q_i.T_K[c][idx] + k_j.T_Q[c][idx]),  out_i = sum_j softmax_j(e_ij) (v_j + sum_c T_V[c][idx]),
 *   idx = clamp(floor(s_i[c] - s_j[c] + rows_c / 2), 0, rows_c - 1)   in fp32. */
int ptv3_swin_attn_fwd(const void* q, const void* k, const void* v, const float* q_table, const float* k_table,
                       const float* v_table, const int32_t* table_offsets_host, int num_axes, const int64_t* n2n,
                       const int32_t* w_start, int num_windows, const float* n_crse, void* out, int64_t n, int heads,
                       int head_dim, int max_tokens, int dtype, void* stream);

/* backward of ptv3_swin_attn_fwd (the reference: SelfAttnAIOFunction.backward of the absent microsoft/Swin3D; checked
 * against torch autograd over the restated forward: parity unpinned).  dout (n, heads, head_dim) in dtype; writes dq, dk, dv
 * (same shape / dtype) and the fp32 gradients of the three concatenated tables (sum(table_offsets) elements each; zeroed
 * here, then accumulated with atomics: their last bits depend on the order of additions). */
int ptv3_swin_attn_bwd(const void* q, const void* k, const void* v, const void* dout, const float* q_table,
                       const float* k_table, const float* v_table, const int32_t* table_offsets_host, int num_axes,
                       const int64_t* n2n, const int32_t* w_start, int num_windows, const float* n_crse, void* dq,
                       void* dk, void* dv, float* dq_table, float* dk_table, float* dv_table, int64_t n, int heads,
                       int head_dim, int max_tokens, int dtype, void* stream);

/* ---- pointops (libs/pointops) ------------------------------------------------------------------
 * Same argument meaning as the reference's extern "C" launchers
 * (libs/pointops/src/knn_query/knn_query_cuda_kernel.h:9-17, grouping/grouping_cuda_kernel.h,
 * interpolation/interpolation_cuda_kernel.h) plus the stream. */
int ptv3_knn_query(int m, int nsample, const float* xyz, const float* new_xyz, const int* offset,
                   const int* new_offset, int b, int* idx, float* dist2, void* stream);
/* k nearest by a walk over the occupied cells of a uniform grid (cells of edge `cell`, the unit of xyz) instead of a
 * scan of the scene: the k smallest by (distance, candidate index), each row ascending in that order.  Against
 * ptv3_knn_query (= the reference's scan) every row holds the same distances and the same neighbours strictly closer than
 * its k-th distance; which of several candidates AT the k-th distance is kept, and the order inside a group of equal
 * distances, are by-products of the reference's heap and differ (lowest indices here).  nsample = 1: identical.
 * qcell (m,4) int32: [scene, cx, cy, cz] of every query, cx = floor(x / cell) - min over the scene set, 0..65535;
 * table / slots: ptv3_subm_build_table over the DISTINCT candidate cells (ncell,4); order (n) int64 + seg_start
 * (ncell+1) int32: the candidates of cell c are order[seg_start[c] .. seg_start[c+1]) (ptv3_argsort_i64 +
 * ptv3_pool_segments of the candidates' cell keys); offset (b) int32: the candidates' cumulative scene ends.  A query
 * whose k-th neighbour is not settled within 8 shells (sparse surroundings, or a scene with fewer than nsample
 * candidates, which pads with idx -1, dist2 1e10 as ptv3_knn_query does) is answered by a scan of its scene. */
int ptv3_knn_query_cells(int m, int nsample, const float* xyz, const float* new_xyz, const int32_t* qcell,
                         const void* table, int64_t slots, const int64_t* order, const int32_t* seg_start,
                         const int* offset, float cell, int* idx, float* dist2, void* stream);
int ptv3_grouping_forward(int m, int nsample, int c, const float* input, const int* idx, float* output,
                          void* stream);
int ptv3_grouping_backward(int m, int nsample, int c, const float* grad_output, const int* idx,
                           float* grad_input, void* stream);
int ptv3_interpolation_forward(int n, int c, int k, const float* input, const int* idx,
                               const float* weight, float* output, void* stream);
int ptv3_interpolation_backward(int n, int c, int k, const float* grad_output, const int* idx,
                                const float* weight, float* grad_input, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* PTV3_HIP_H */
