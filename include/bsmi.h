/* bsmi.h -- C ABI of libbsmi.so, the MI355X (gfx950) engine behind the
 * `bs predict` / `bs segment --ws` hot path of ucsdmanorlab/bootstrapper.
 *
 * Plain pointers and sizes only; no torch / numpy types.  All functions return
 * 0 on success and a negative code on failure; the message for the calling
 * thread is available from bsmi_last_error().  No exception crosses this
 * boundary.  Pointers suffixed _dev are device (HBM) pointers on the handle's
 * device, _host are host pointers.  `stream` is a hipStream_t passed as void*
 * (NULL = the null stream); calls are asynchronous on it unless stated.
 *
 * Every entry point names the reference interface it replaces (paths relative
 * to /root/reference/bootstrapper).
 */
#ifndef BSMI_H
#define BSMI_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BSMI_MAX_LEVELS 8
#define BSMI_MAX_CONVS 4
#define BSMI_MAX_HEADS 4
#define BSMI_NAME_LEN 32

#define BSMI_OK 0
#define BSMI_ERR_INVALID (-1)   /* bad argument / shape the network cannot take */
#define BSMI_ERR_HIP (-2)       /* HIP runtime error                            */
#define BSMI_ERR_STATE (-3)     /* call order (e.g. forward before finalize)    */
#define BSMI_ERR_MISSING (-4)   /* missing / unexpected state-dict key          */
#define BSMI_ERR_OVERFLOW (-5)  /* a fixed-capacity device structure overflowed */

/* arithmetic type of the U-Net contraction */
#define BSMI_PREC_F32 0   /* f32 MFMA (exact f32 products, f32 accumulate): parity mode */
#define BSMI_PREC_BF16 1  /* bf16 MFMA operands, f32 accumulate: throughput mode        */
#define BSMI_PREC_BF16X3 2 /* split bf16: every f32 operand x = hi + lo (two bf16), products hi*hi + lo*hi + hi*lo
                              on the bf16 MFMA, f32 accumulate; activations stored as (hi, lo) planes.  Within the
                              1e-4 parity gate of the fp32 reference at 3 MFMAs per product                        */
#define BSMI_NUM_PREC 3

/* dtype of the raw input handed to bsmi_unet_forward */
#define BSMI_RAW_U8 0     /* uint8 [Cin][D][H][W]; normalised on device as u8/255*2-1    */
#define BSMI_RAW_F32 1    /* float [Cin][D][H][W]; already normalised                    */
#define BSMI_RAW_U8_UNIT 2 /* uint8 [Cin][D][H][W]; normalised on device as u8/255: predictions
                              fed to a second-stage net (models/3d_affs_from_2d_mtlsd/predict.py:163-164) */

const char *bsmi_last_error(void);
int bsmi_version(void);

/* ------------------------------------------------------------------------ */
/* 3-D U-Net (reference: models/3d_affs/unet.py:226-478 UNet, :7-76 ConvPass,
 * :79-106 Downsample, :109-223 Upsample; models/3d_affs/model.py:28-64 Model;
 * models/3d_mtlsd/model.py:28-68).                                            */

/* Mirrors the keys of net_config.json that Model() consumes
 * (models/3d_affs/model.py:10-25, net_config.json) plus the head list the
 * Model class hard-codes (model.py:54-56; mtlsd model.py:54-59).             */
typedef struct bsmi_unet_config {
  int32_t in_channels;
  int32_t num_fmaps;
  int32_t fmap_inc_factor;
  int32_t num_levels; /* len(downsample_factors) + 1 */
  int32_t downsample_factors[BSMI_MAX_LEVELS][3];
  int32_t n_convs_down[BSMI_MAX_LEVELS];
  int32_t kernel_size_down[BSMI_MAX_LEVELS][BSMI_MAX_CONVS][3];
  int32_t n_convs_up[BSMI_MAX_LEVELS];
  int32_t kernel_size_up[BSMI_MAX_LEVELS][BSMI_MAX_CONVS][3];
  int32_t num_heads;
  char head_name[BSMI_MAX_HEADS][BSMI_NAME_LEN]; /* state-dict prefix, forward() return order */
  int32_t head_dims[BSMI_MAX_HEADS];
  int32_t num_fmaps_out; /* channels of the last right-side ConvPass (unet.py:239,344,426); 0 = num_fmaps.
                            Set by the second-stage nets (net_config.json of the models/3d_affs_from_... setups) */
} bsmi_unet_config;

typedef struct bsmi_unet bsmi_unet;

/* replaces Model.__init__ (model.py:30-56) on `device` */
int bsmi_unet_create(const bsmi_unet_config *cfg, int device, bsmi_unet **out);
int bsmi_unet_destroy(bsmi_unet *h);

/* replaces Model.load_state_dict (models/3d_affs/predict.py:103-107): one call per
 * state-dict entry, `key` WITHOUT the Lightning "model." prefix, data OIDHW fp32. */
int bsmi_unet_load_weight(bsmi_unet *h, const char *key, const float *data_host,
                          const int64_t *shape, int ndim);
/* strict check (every expected key loaded) + pack/upload weights for `precision`;
 * may be called once per precision. */
int bsmi_unet_finalize(bsmi_unet *h, int precision);

/* valid-conv shape arithmetic (unet.py:96-104 divisibility, :147-201 crop_to_factor) */
int bsmi_unet_output_shape(bsmi_unet *h, const int64_t in_shape[3], int64_t out_shape[3]);
/* algorithmic multiply-add count x2 for one forward at in_shape (SURVEY 8a table:
 * residual 1x1x1 counted on the cropped extent) */
int bsmi_unet_flops(bsmi_unet *h, const int64_t in_shape[3], double *flops);

/* replaces Model.forward (model.py:58-64) + the predict worker's arithmetic around
 * it (models/3d_affs/predict.py:145-154): raw -> heads.  For each head i either
 * output pointer may be NULL:
 *   out_f32_dev[i]: float  [head_dims[i]][d][h][w]  sigmoid outputs
 *   out_u8_dev[i] : uint8  [head_dims[i]][d][h][w]  (uint8)(sigmoid*255), truncation
 * Device workspace for the given in_shape is allocated on first use and cached. */
int bsmi_unet_forward(bsmi_unet *h, int precision, const void *raw_dev, int raw_dtype,
                      const int64_t in_shape[3], float *const *out_f32_dev,
                      uint8_t *const *out_u8_dev, void *stream);

/* ---- training (reference models/3d_affs/train.py:152-159, model.py:67-92; fp32 like the reference) ------------
 * begin: after bsmi_unet_finalize(h, BSMI_PREC_F32); builds the backward plan for in_shape and moves the
 *   parameters into a flat fp32 device buffer (sorted state_dict keys, each padded to 4 floats).
 * forward_backward: raw_dev float32 [Cin][D][H][W] (already normalised, as the training pipeline delivers it);
 *   targets_dev[i] / weights_dev[i]: float32 [head_dims[i]][d][h][w] per head, in Model.forward order; the loss is
 *   the sum over heads of WeightedMSELoss; gradients of all parameters land in the flat gradient buffer
 *   (zeroed first).  loss_host (optional) synchronises the stream.
 * buffers: the flat parameter / gradient device buffers (the gradient buffer is what data-parallel training
 *   all-reduces); param_info: offset and count of one state_dict key inside them.
 * adam_step: torch.optim.Adam semantics (no weight decay); grad_scale multiplies the gradients first (1/world_size
 *   after a summing all-reduce); the packed weight images of all launches are rewritten from the new parameters.
 * read_param: what = 0 parameter, 1 gradient, 2 / 3 Adam moments -> host.  end: frees the training state; the trained
 *   parameters stay the handle's weights (re-finalize BSMI_PREC_BF16 before predicting in bf16).
 * set_arithmetic (before begin): 1 (default) = the convolutions of the step -- forward, input gradients, weight gradients --
 *   multiply as split-bf16 (every f32 operand = bf16 hi + bf16 lo, hi*hi + lo*hi + hi*lo accumulated in f32: 2^-17
 *   relative per product, what BSMI_PREC_BF16X3 is to inference) while every tensor, the loss, the gradient buffer and
 *   Adam stay fp32; 0 = exact f32 MFMA throughout (half the speed). */
int bsmi_unet_train_set_arithmetic(bsmi_unet *h, int split_bf16);
/* set_deterministic (any time; default 0): 1 = every reduction of the step that float atomics would order by chance -- the loss
 *   sums, the head's and the biases' gradients, the weight gradients of launches cut into line ranges, the transposed
 *   interpolation -- is done as per-workgroup partial results added in index order: two runs of a step on the same inputs give
 *   the same bits (gradients, loss, parameters after Adam), like the reference's CPU path.  Costs about an eighth of the step
 *   (21.2 against 18.7 ms on the (32,196,196) block). */
int bsmi_unet_train_set_deterministic(bsmi_unet *h, int on);
int bsmi_unet_train_begin(bsmi_unet *h, const int64_t in_shape[3]);
int bsmi_unet_train_forward_backward(bsmi_unet *h, const float *raw_dev, const float *const *targets_dev,
                                     const float *const *weights_dev, float *loss_host, void *stream);
int bsmi_unet_train_num_params(bsmi_unet *h, uint64_t *count);
int bsmi_unet_train_buffers(bsmi_unet *h, float **params_dev, float **grads_dev);
int bsmi_unet_train_param_info(bsmi_unet *h, const char *key, uint64_t *offset, uint64_t *count);
int bsmi_unet_train_adam_step(bsmi_unet *h, float lr, float beta1, float beta2, float eps, float grad_scale,
                              void *stream);
int bsmi_unet_train_read_param(bsmi_unet *h, const char *key, int what, float *host_out);
/* last_loss: the loss of the last forward_backward (synchronises `stream`), for callers that passed loss_host = NULL to
 *   keep the step asynchronous.
 * grad_groups: the flat buffers cut into the ranges whose gradients become final together (one per ConvPass / head), in
 *   the order the backward pass finishes them; wait_grad_group makes `stream` wait until the last forward_backward has
 *   finished group g -- a data-parallel caller reduces group after group on a side stream while the backward pass
 *   still runs (the implicit DDP of training.py:125-133 overlaps bucket all-reduces the same way).
 * write_param (what = 2 / 3) and step_count restore the Adam moments and step of a checkpoint (Lightning resumes them,
 *   training.py:131-137 ckpt_path); step_count(set_to < 0) only reads. */
int bsmi_unet_train_last_loss(bsmi_unet *h, float *loss_host, void *stream);
/* the sigmoid outputs of head `head` from the last forward_backward, float [dims][d][h][w] on the device (what the
 * reference's training_step returns as pred_<head> for the snapshot callback, models/3d_mtlsd/train.py:183-187) */
int bsmi_unet_train_prediction(bsmi_unet *h, int head, float **out_dev, uint64_t *count);
int bsmi_unet_train_grad_groups(bsmi_unet *h, int max_n, int *n, uint64_t *offsets, uint64_t *counts);
int bsmi_unet_train_wait_grad_group(bsmi_unet *h, int group, void *stream);
int bsmi_unet_train_write_param(bsmi_unet *h, const char *key, int what, const float *host_in);
int bsmi_unet_train_step_count(bsmi_unet *h, int set_to, int *value);
int bsmi_unet_train_end(bsmi_unet *h);

/* Affinity training targets of one sample, on the device (reference models/3d_affs/train.py:127-139:
 * gp.GrowBoundary(labels, mask=unlabelled, steps, only_xy) -> gp.AddAffinities(neighborhood) ->
 * gp.BalanceLabels; the erosion as in gp/custom_grow_boundary.py:71-110 with a fixed step count).
 *   labels_dev      int64 [D][H][W], 0 = background; overwritten with the grown-boundary labels
 *   unlabelled_dev  uint8 [D][H][W] or NULL: 1 where the ground truth is known (CreateMask), 0 = unknown
 *   neighborhood    n offsets (z, y, x), host pointer (net_config "neighborhood")
 *   grow_steps      voxels of boundary to grow (net_config "grow_boundary"); a voxel keeps its label when every
 *                   voxel within that L1 distance (in its section if only_xy) has the same label, is unknown, or
 *                   lies outside the block (binary_erosion(iterations=steps, border_value=1))
 *   affs_dev        float [n][D][H][W]: 1 where p and p + offset carry the same non-zero label
 *   weights_dev     float [n][D][H][W]: affinity mask (both voxels inside the block, p known) scaled by
 *                   1 / (2 f) for positives and 1 / (2 (1 - f)) for negatives, f = masked positive fraction
 *                   clipped to [clip_min, clip_max] (BalanceLabels, slab = whole sample)                       */
int bsmi_train_affinity_targets(int device, int64_t *labels_dev, const uint8_t *unlabelled_dev,
                                const int64_t shape[3], const int32_t *neighborhood, int n, int grow_steps,
                                int only_xy, float clip_min, float clip_max, float *affs_dev,
                                float *weights_dev, void *stream);

/* Local-shape-descriptor training targets of one sample, on the device (reference models/3d_mtlsd/train.py:134-141:
 * AddLocalShapeDescriptor(labels, gt_lsds, unlabelled, lsds_mask, sigma, downsample) of the lsd package [EXT]; restated
 * in oracle/lsd_ref.py).  3-D descriptors, 10 channels: mean offset (z, y, x), variances, Pearson coefficients
 * (zy, zx, yx), size -- each in [0, 1].
 *   labels_dev   int64 [D][H][W]: the label crop grown by the window context (3 sigma; zeros where the volume ends)
 *   unlabelled_dev  uint8 [D][H][W] or NULL: 1 where the ground truth is known
 *   roi_offset / roi_shape  the output block inside that array (all multiples of `downsample`, like the lsd package asks)
 *   sigma, voxel_size  world units (net_config "sigma" is one number: the same for the three axes)
 *   lsds_dev     float [10][d][h][w];  weights_dev  float [10][d][h][w] or NULL: 1 on labelled, known voxels        */
int bsmi_train_lsd_targets(int device, const int64_t *labels_dev, const uint8_t *unlabelled_dev, const int64_t shape[3],
                           const int64_t roi_offset[3], const int64_t roi_shape[3], const float sigma[3],
                           const float voxel_size[3], int downsample, float *lsds_dev, float *weights_dev, void *stream);

/* Number of CUs the stream bsmi_unet_forward is called on may use (a multiple of 8; -1 restores the
 * default = all CUs of the device, 0 disables the persistent launches).  The big-tile conv layers run
 * as that many persistent workgroups (conv_igemm.hip); set it when the stream carries a CU mask. */
int bsmi_unet_set_persistent_grid(bsmi_unet *h, int n_cus);

/* HIP stream restricted to the CUs of `cu_mask` (hipExtStreamCreateWithCUMask; on MI355X bit i selects a
 * CU of XCD i % 8, so the first 8k bits give k CUs in every XCD).  The block pipeline keeps its
 * latency-bound segmentation kernels and the U-Net on disjoint CU sets this way. */
int bsmi_stream_create_cu_mask(int device, const uint32_t *cu_mask, int n_words, void **stream_out);
int bsmi_stream_destroy(int device, void *stream);

/* Per-launch timing of the forward pass with HIP events recorded on the caller's stream
 * (bench.py's roofline leg).  on = N > 0: every Nth bsmi_unet_forward (N = 1: every one)
 * brackets each launch with events; on = 0 switches the timing off.
 * bsmi_unet_profile_read synchronises on them and returns, for the last timed forward, the launch type (0 input, 1 implicit-GEMM conv, 2 max-pool, 3 upsample+crop,
 * 4 head), its duration in ms and its algorithmic FLOPs. */
int bsmi_unet_profile_enable(bsmi_unet *h, int on);
int bsmi_unet_profile_read(bsmi_unet *h, int max_n, int *n, int32_t *types, double *ms,
                           double *flops);
/* totals over every profiled forward since the last reset, indexed by launch type */
int bsmi_unet_profile_totals(bsmi_unet *h, double ms_by_type[5], double flops_by_type[5],
                             int64_t launches_by_type[5], int reset);
/* FLOPs the matrix pipe was actually given by the profiled convolution launches since the last reset (tile padding, every
 * batch of a Winograd stage, three bf16 products per product of the split mode): the algorithmic count of
 * bsmi_unet_profile_totals says how fast the layer was computed, this one how busy the MFMA units were -- a Winograd stage
 * computes a layer with fewer multiplies than the algorithmic count has. */
int bsmi_unet_profile_executed(bsmi_unet *h, double *executed_flops, int reset);
/* Development aid: `blocks` one-wave workgroups that fill `lds_bytes` of LDS with a pattern, idle for `spins` x 127 x 64 clocks
 * and check it; *mismatches_dev (device uint64, caller-zeroed) counts words that changed.  Run beside an engine on another
 * stream: does a co-resident kernel write outside its own LDS allocation? */
int bsmi_debug_lds_canary(int lds_bytes, int blocks, int spins, unsigned long long *mismatches_dev, void *stream);
/* Development aid: with BSMI_GUARD_MB=<n> in the environment every device allocation of the network engine, the training state and
 * the segmentation engines lies between
 * two n-MiB zones of 0xFF bytes (csrc/dev_guard.h): a read past a buffer that reaches a result turns it into NaNs, and this
 * call counts the zones something WROTE into (0 = intact, each hit reported on stderr; -1 = HIP error; 0 when unset). */
int bsmi_debug_check_guards(void);

/* Development aid: the output tensor of launch `step` of the last forward (launch order as in
 * bsmi_unet_profile_read), as float32 channels-last [D][H][W][C] on the host, whatever the precision mode stores
 * (f32, bf16, or the hi + lo planes of the split mode).  shape_out = {D, H, W, C}; host_out may be NULL to query
 * the shape.  what: 0 = the value, 1 / 2 = only the hi / lo plane of the split mode.  Synchronises the device. */
int bsmi_unet_debug_activation(bsmi_unet *h, int step, int what, int64_t shape_out[4], float *host_out,
                               uint64_t capacity);

/* Reflect-padded block extraction (gp.Pad(raw, None, mode="reflect") +
 * ArraySource ROI read, models/3d_affs/predict.py:145-148): copies the window
 * [offset, offset+block_shape) of vol (uint8 [D][H][W]) into block, mirroring
 * coordinates outside the volume (numpy 'reflect': edge voxel not repeated). */
int bsmi_extract_block_reflect_u8(const uint8_t *vol_dev, const int64_t vol_shape[3],
                                  const int64_t offset[3], const int64_t block_shape[3],
                                  uint8_t *block_dev, void *stream);

/* ------------------------------------------------------------------------ */
/* Seeded watershed fragments (reference: post/ws.py:38-112
 * watershed_from_affinities, :8-35 watershed_from_boundary_distance).
 * affs_dev: uint8 [3][D][H][W] (z,y,x nearest-neighbour affinities).
 * frags_dev: uint64 [D][H][W].  max_id_dev: uint64[1] (ws.py return value n+id_offset).
 * Only fragments_in_xy != 0 and max_affinity_value = 255 are implemented.        */
typedef struct bsmi_seg bsmi_seg;
int bsmi_seg_create(int device, const int64_t max_shape[3], bsmi_seg **out);
int bsmi_seg_destroy(bsmi_seg *h);

int bsmi_ws_fragments_u8(bsmi_seg *h, const uint8_t *affs_dev, const int64_t shape[3],
                         int fragments_in_xy, int min_seed_distance, uint64_t *frags_dev,
                         uint64_t *max_id_dev, void *stream);

/* The same with `return_seeds` (ws.py:42,105-110): seeds_dev (uint64 [D][H][W], may be NULL) receives the labelled
 * maxima before they meet the mask, with the same id offsets as the fragments they grow into. */
int bsmi_ws_fragments_seeds_u8(bsmi_seg *h, const uint8_t *affs_dev, const int64_t shape[3],
                               int fragments_in_xy, int min_seed_distance, uint64_t *frags_dev,
                               uint64_t *max_id_dev, uint64_t *seeds_dev, void *stream);

/* Mean-affinity hierarchical agglomeration (reference call site
 * post/watershed.py:333-338 waterz.agglomerate(affs, thresholds, fragments,
 * "OneMinus<MeanAffinity<RegionGraphType, ScoreValue>>"); algorithm restated in
 * oracle/seg_ref.c).  For each threshold t (ascending) writes
 * segs_dev + t*D*H*W: uint64 [D][H][W].  frags_dev is not modified.               */
int bsmi_agglomerate_mean_u8(bsmi_seg *h, const uint8_t *affs_dev, const uint64_t *frags_dev,
                             const int64_t shape[3], const float *thresholds_host,
                             int n_thresholds, uint64_t *segs_dev, void *stream);

/* The same hierarchical agglomeration with a histogram-quantile scorer (reference post/watershed.py:230-243:
 * merge_function "hist_quant_<Q>[_initmax]" = "OneMinus<HistogramQuantileAffinity<RegionGraphType, Q, ScoreValue, 256,
 * init_with_max>>"; waterz call site post/watershed.py:333-338).  The region graph and the 256-bin histogram of every edge's
 * affinities are built on the device, the merge loop runs on the host (as waterz's does), the relabel on the device.
 * Synchronises the stream.  Outputs as bsmi_agglomerate_mean_u8.                                                     */
int bsmi_agglomerate_hist_u8(bsmi_seg *h, const uint8_t *affs_dev, const uint64_t *frags_dev,
                             const int64_t shape[3], const float *thresholds_host, int n_thresholds,
                             int quantile, int init_with_max, uint64_t *segs_dev, void *stream);

/* The host half of bsmi_agglomerate_hist_u8 on a region graph given by the caller (no GPU involved): n_nodes nodes, edges
 * (edge_u[e] < edge_v[e], each pair once) with their 256-bin affinity histograms hist [n_edges][256] (modified); for every
 * threshold (ascending) roots_out + t * n_nodes receives the smallest node of each node's cluster after mergeUntil.     */
int bsmi_agglomerate_hist_graph(uint32_t n_nodes, uint32_t n_edges, const uint32_t *edge_u, const uint32_t *edge_v,
                                uint32_t *hist, int quantile, int init_with_max, const float *thresholds,
                                int n_thresholds, uint32_t *roots_out);

/* Blockwise fragment post-processing (reference post/blockwise/watershed_frags.py:148-156 filter_avg_fragments,
 * :188-192 remove_small_objects, :221-224 crop to the write ROI + skimage.measure.label + global id offset).
 * frags_dev (the read-ROI fragments of bsmi_ws_fragments_u8) is filtered IN PLACE: a fragment is removed
 * if the mean of its 3-channel average affinity (u8/255) is < filter_value (skipped when <= 0) or if it has
 * fewer than min_size voxels (skipped when <= 0).  The crop [crop_offset, +crop_shape) is then relabelled:
 * connected components of equal value under 26-connectivity, numbered id_offset+1.. in raster order of
 * their first voxel; *num_labels_dev receives the count. */
int bsmi_frag_postprocess_u8(bsmi_seg *h, const uint8_t *affs_dev, uint64_t *frags_dev,
                             const int64_t shape[3], double filter_value, int64_t min_size,
                             const int64_t crop_offset[3], const int64_t crop_shape[3],
                             uint64_t id_offset, uint64_t *out_dev, uint64_t *num_labels_dev,
                             void *stream);

/* RAG node attributes of a labelled block (watershed_frags.py:230-246): for labels id_offset+1 ..
 * id_offset+num, size_dev[k] = voxel count and sums_dev[3k..3k+2] = sums of the z, y, x voxel indices
 * (centre of mass = sums / size). */
int bsmi_label_stats(bsmi_seg *h, const uint64_t *labels_dev, const int64_t shape[3], uint64_t id_offset,
                     uint64_t num, uint64_t *size_dev, uint64_t *sums_dev, void *stream);

/* Per-block RAG edge scoring (reference post/blockwise/waterz_agglom.py:106-170): dense relabel of the
 * fragment ids in ascending order (:116-120), waterz.agglomerate(thresholds=[0, threshold],
 * discretize_queue, return_merge_history, return_region_graph) (:131-139), MergeTree replay and per-edge
 * merge score (:153-170; post/merge_tree.py:5-113).  frags_dev may hold arbitrary 64-bit ids (< 2^64-2).
 * Outputs: edges_dev[2e], edges_dev[2e+1] = ids (u < v) of initial RAG edge e, in ascending (u, v) order;
 * scores_dev[e] = score of the merge that first joined u and v, NaN if they never merge below `threshold`;
 * merges_dev[2i], [2i+1] = (surviving id, absorbed id) and merge_scores_dev[i] of merge i (both optional);
 * counts_dev[0..2] = number of edges, merges, nodes.  bsmi_seg_status reports an edge_capacity overflow. */
int bsmi_rag_merge_scores_u8(bsmi_seg *h, const uint8_t *affs_dev, const uint64_t *frags_dev,
                             const int64_t shape[3], float threshold, int discretize_queue,
                             uint64_t *edges_dev, float *scores_dev, uint64_t edge_capacity,
                             uint64_t *merges_dev, float *merge_scores_dev, uint64_t *counts_dev,
                             void *stream);

/* The region graph of a block without the merge loop, for bsmi_rag_merge_scores_host: edges_dev [edge_capacity][2] fragment id
 * pairs (smaller id first) in the order of the device's edge table (NOT sorted, and not the same from run to run:
 * bsmi_rag_merge_scores_host sorts), sums_dev [edge_capacity] affinity sums and pair_counts_dev [edge_capacity] voxel-pair counts
 * of their faces, counts_dev[0..2] = number of edges, 0, nodes.  bsmi_seg_status reports an edge_capacity overflow (counts_dev[0]
 * then says how many entries the block needs).  Four launches; the ordered form behind bsmi_rag_merge_scores_u8 is sixty. */
int bsmi_rag_graph_u8(bsmi_seg *h, const uint8_t *affs_dev, const uint64_t *frags_dev, const int64_t shape[3],
                      uint64_t *edges_dev, uint64_t *sums_dev, uint32_t *pair_counts_dev, uint64_t edge_capacity,
                      uint64_t *counts_dev, void *stream);
/* waterz_agglom.py:106-170 on the host for n_graphs graphs of bsmi_rag_graph_u8 (host copies) side by side on up to n_threads
 * threads (<= 0: 24), largest first.  Every graph is first brought into ascending (id, id) order IN PLACE (edges, sums and
 * pair_counts permuted together), then agglomerated to `threshold` with OneMinus<MeanAffinity> and a `discretize_queue`-bin queue;
 * every edge's score = the score at which its two regions merged (NaN: never) -> scores[g][e], e in the sorted order.  Same
 * results as bsmi_rag_merge_scores_u8. */
int bsmi_rag_merge_scores_host(int n_graphs, const uint64_t *n_edges, uint64_t *const *edges, uint64_t *const *sums,
                               uint32_t *const *pair_counts, float threshold, int discretize_queue, float *const *scores,
                               int n_threads);
/* The same with the queue's bin rule as an argument.  waterz's `discretize_queue` binning (reference
 * post/blockwise/waterz_agglom.py:136; waterz is an unpinned git dependency that is not in /root/reference) cannot be checked
 * in this repository, so the rule is a documented choice (segment config key `queue_bins_formula`):
 * BSMI_QUEUE_BINS_N_MINUS_1 (default, what the oracle and the device loop do) bin = (int)(score * (N - 1));
 * BSMI_QUEUE_BINS_N bin = min(N - 1, (int)(score * N)).  tools/gen_goldens_waterz.py produces the vectors that decide. */
#define BSMI_QUEUE_BINS_N_MINUS_1 0
#define BSMI_QUEUE_BINS_N 1
int bsmi_rag_merge_scores_host_rule(int n_graphs, const uint64_t *n_edges, uint64_t *const *edges, uint64_t *const *sums,
                                    uint32_t *const *pair_counts, float threshold, int discretize_queue, int bin_rule,
                                    float *const *scores, int n_threads);

/* Epsilon agglomeration of a block's fragments IN PLACE (reference post/blockwise/watershed_frags.py:158-177:
 * waterz.agglomerate(thresholds=[epsilon], fragments, "OneMinus<MeanAffinity>", discretize_queue=256), result written
 * back into the fragments): every fragment takes the id of its cluster, the smallest id among the fragments merged
 * into it.  Same region graph and merge loop as bsmi_rag_merge_scores_u8. */
int bsmi_rag_agglomerate_u8(bsmi_seg *h, const uint8_t *affs_dev, uint64_t *frags_dev, const int64_t shape[3],
                            float threshold, int discretize_queue, void *stream);

/* Affinity sum (uint8 units) and voxel-pair count of every initial RAG edge of the last bsmi_rag_merge_scores_u8 call
 * on this handle, in the order of its edges_dev: what waterz's region graph {u, v, score} is computed from
 * (score = 1 - sum / (255 count)); also lets a caller contract the graph along the merge history. */
int bsmi_rag_edge_stats(bsmi_seg *h, uint64_t *sums_dev, uint64_t *counts_dev, uint64_t capacity, void *stream);

/* LUT relabel (volara Relabel + LUT, post/watershed.py:187-202): out[p] = vals[k] where keys[k] == in[p]
 * (keys ascending); 0 stays 0; ids without a key are copied.  in_dev == out_dev is allowed. */
int bsmi_lut_relabel(int device, const uint64_t *in_dev, uint64_t n, const uint64_t *keys_dev,
                     const uint64_t *vals_dev, uint64_t m, uint64_t *out_dev, void *stream);
/* The same with n_columns value columns at once (one segmentation per threshold): vals_dev [n_columns][m], out_dev
 * [n_columns][n]; one look-up per run of equal ids serves every column.  out_dev must not overlap in_dev. */
int bsmi_lut_relabel_multi(int device, const uint64_t *in_dev, uint64_t n, const uint64_t *keys_dev,
                           const uint64_t *vals_dev, uint64_t m, int n_columns, uint64_t *out_dev, void *stream);

/* Global thresholded connected components of the scored RAG on the HOST (plain host pointers; reference
 * post/watershed.py:182 calls funlib.segment.graphs.impl.connected_components, a host C++ routine).
 * nodes strictly ascending; an edge joins its endpoints when score <= threshold; components[i] = smallest
 * node id of node i's component.  Edges naming unknown nodes are ignored. */
int bsmi_connected_components(const uint64_t *nodes, uint64_t n, const uint64_t *edges,
                              const float *scores, uint64_t m, float threshold, uint64_t *components);
/* The same for several thresholds in one pass (reference post/watershed.py:177-186 loops over the thresholds): the node
 * look-up of the edges is done once, on a few host threads, and the unions of a lower threshold carry over to the higher
 * ones.  components: [n_thresholds][n], row k for thresholds[k]. */
int bsmi_connected_components_multi(const uint64_t *nodes, uint64_t n, const uint64_t *edges,
                                    const float *scores, uint64_t m, const float *thresholds,
                                    int n_thresholds, uint64_t *components);

/* Thresholded-affinity connected components (`bs segment --cc`; reference post/cc.py:7-74 called from
 * post/connected_components.py:77-80, debris removal :97-101).  Voxel p is linked with its +z / +y / +x neighbour when
 * affs[d][p] > cut (cut = the uint8 value equivalent to the reference's float threshold: the largest v with
 * !(v / 255.0f > threshold)).  frags_dev: labels 1.. in raster order of each component's first voxel; seg_dev
 * (optional): the same with components of fewer than min_size voxels removed; *num_labels_dev = component count. */
int bsmi_cc_affs_u8(bsmi_seg *h, const uint8_t *affs_dev, const int64_t shape[3], int cut, int64_t min_size,
                    uint64_t *frags_dev, uint64_t *seg_dev, uint64_t *num_labels_dev, void *stream);

/* ---- mutex watershed (`bs segment --mws`) ------------------------------------------------------------------
 * Replaces mwatershed.agglom as the reference calls it (post/mws.py:51-56; the package is third party and absent:
 * restated from Wolf et al., "The Mutex Watershed", parity unpinned).  affs_dev: f64 [n_offsets][D][H][W] on the
 * device, already shifted (w > 0 attractive, w < 0 repulsive, 0 / NaN: no edge); offsets / strides: host
 * int32 [n_offsets][3] (strides NULL = every voxel; random_seed != 0 = the "randomized_strides" form: an edge is kept
 * with probability 1 / prod(stride)).  Edges in order of |w| descending, ties (channel, voxel) ascending.
 * labels_dev: u64 [D][H][W], 1 + smallest voxel index of the cluster.  Synchronises `stream` (the sweep over the
 * sorted edges runs on the host: it is sequential by definition). */
int bsmi_mws_agglom_f64(int device, const double *affs_dev, int n_offsets, const int32_t *offsets,
                        const int32_t *strides, uint32_t random_seed, const int64_t shape[3],
                        uint64_t *labels_dev, void *stream);

/* Mutex watershed on a graph, HOST pointers (volara GraphMWS -> mwatershed.cluster, reference
 * post/watershed_mutex.py:155-161): edges u64 [m][2] of node indices < n_nodes, scores f64 [m] (sign as above),
 * labels_out[i] = 1 + smallest node index of node i's cluster. */
int bsmi_mws_cluster(uint64_t n_nodes, const uint64_t *edges, const double *scores, uint64_t m,
                     uint64_t *labels_out);

/* Affinity statistics between adjacent fragments over a neighbourhood (volara AffAgglom, reference
 * post/watershed_mutex.py:143-153, scores={"zyx_aff": neighborhood}).  affs_dev: u8 [n_offsets][D][H][W]; dense_dev:
 * u64 [D][H][W] of dense fragment indices 1..n < 2^32 (0 = background); for every unordered pair of different
 * fragments joined by some (p, p + offset_k): pairs_out[i] = {smaller, larger} (HOST, ascending), sums_out[i] = sum
 * of the affinity bytes, counts_out[i] = number of such voxel pairs.  *n_pairs is always set; BSMI_ERR_INVALID if
 * it exceeds `capacity`. */
int bsmi_frag_pair_affinity_u8(int device, const uint8_t *affs_dev, int n_offsets, const int32_t *offsets,
                               const uint64_t *dense_dev, const int64_t shape[3], uint64_t capacity,
                               uint64_t *pairs_out, uint64_t *sums_out, uint64_t *counts_out,
                               uint64_t *n_pairs, void *stream);

/* Label table of a block (`bs refine` statistics: reference refine.py:98-109 `_global_sizes`, :228-250 z extents):
 * the distinct non-zero ids of labels_dev in ascending order with their voxel counts and first / last z slice
 * (z0 = global index of the block's first slice).  *n_dev = number of ids (<= capacity, else bsmi_seg_status
 * reports an overflow).  The filters themselves mask / remap through bsmi_lut_relabel. */
int bsmi_label_table_u64(bsmi_seg *h, const uint64_t *labels_dev, const int64_t shape[3], int64_t z0,
                         uint64_t *ids_dev, uint64_t *counts_dev, int32_t *zmin_dev, int32_t *zmax_dev,
                         uint64_t capacity, uint64_t *n_dev, void *stream);

/* fragments_in_xy = 0 (reference post/ws.py:98-110) floods the block from ONE priority queue.  on = 1: that flood runs on the
 * host (csrc/flood_host.cpp: 0.16 s per 128^3 block instead of 12.8 s for the device's single-wave replay) and
 * bsmi_ws_fragments_seeds_u8 returns when the fragments are written (the Python engines switch it on); 0 (the handle's
 * default): the device loop, asynchronous like every other call. */
int bsmi_seg_set_host_flood(bsmi_seg *h, int on);

/* status of the asynchronous seg calls on this handle since the previous bsmi_seg_status (an overflow of any of
 * them is remembered on the device until it is read here; synchronises `stream`): BSMI_OK or BSMI_ERR_OVERFLOW.
 * After an overflow the outputs of that call are undefined. */
int bsmi_seg_status(bsmi_seg *h, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* BSMI_H */
