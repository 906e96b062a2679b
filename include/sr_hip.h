/*
 * sr_hip.h -- C ABI of the MI355X-native tile -> blend -> assess engine (libsrhip.so).
 *
 * This is the drop-in boundary: plain pointers and sizes, no torch / numpy types.  Each entry
 * point names the reference interface it replaces (paths relative to the reference repo).
 * Host-side mirrors of the reference's Python classes (tiling_module.TilingModule,
 * blending_module.BlendingModule, quality_assessment_module.QualityAssessmentModule,
 * main.SuperResolutionPipeline) bind these through ctypes; INTEGRATION.md shows the stub.
 *
 * Conventions
 *  - every function returns an int status (SR_OK == 0, negative = error); the message of the
 *    last error on the calling thread is sr_last_error().
 *  - "d_" pointers are device (HBM) addresses on the context's GPU, "h_" pointers are host.
 *    Small descriptor arrays (tile rectangles etc.) are always host pointers.
 *  - images are row-major, channel-interleaved (HWC) exactly like the reference's ndarrays;
 *    strides are in BYTES.
 *  - all device work is enqueued on the context's stream (sr_ctx_create makes one;
 *    sr_ctx_create_on_stream adopts the caller's, e.g. torch's current stream).  Functions
 *    that return a value to the host synchronise that stream; the others do not.
 *  - a context may be used from several threads: calls on one context are serialised by an
 *    internal mutex (the reference's ParallelBlender calls laplacian_fusion from a thread
 *    pool on one object, blending_module.py:1668-1701).
 *  - the library never keeps a host pointer after a call returns.
 */
#ifndef SR_HIP_H
#define SR_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SR_API __attribute__((visibility("default")))

enum sr_status {
    SR_OK = 0,
    SR_ERR_INVALID_ARG = -1, /* -> ValueError in the Python mirror                    */
    SR_ERR_SHAPE = -2,       /* shape / size mismatch (cv2.error in the reference)    */
    SR_ERR_OOM = -3,
    SR_ERR_HIP = -4,         /* any HIP runtime failure, incl. "no device"            */
    SR_ERR_COMM = -5,
    SR_ERR_UNSUPPORTED = -6
};

/* tiling_module.PaddingMode (tiling_module.py:40-45) */
enum sr_pad_mode { SR_PAD_MIRROR = 0, SR_PAD_REPLICATE = 1, SR_PAD_REFLECT = 2, SR_PAD_CONSTANT = 3 };
/* blending_module.WeightType (blending_module.py:52-56) */
enum sr_weight_type { SR_W_LINEAR = 0, SR_W_COSINE = 1, SR_W_SIGMOID = 2,
                      /* every weight 1: what BlendingModule.feather_blend's distance-transform weights evaluate to
                       * (blending_module.py:1313-1337: cv2.distanceTransform of an all-ones mask; see sr_weighted_blend) */
                      SR_W_ONES = 3 };
/* calculate_ssim branches (quality_assessment_module.py:365-417) */
enum sr_ssim_mode { SR_SSIM_UNIFORM7 = 0, SR_SSIM_GAUSS11 = 1, SR_SSIM_SIMPLE = 2 };
enum sr_dtype { SR_U8 = 0, SR_F32 = 1, SR_F64 = 2 };   /* SR_F64: sr_ssim_float only */

typedef struct sr_ctx sr_ctx;
typedef struct sr_blend_plan sr_blend_plan;

/* ---- library / context ------------------------------------------------------------- */
SR_API int sr_version(void);
/* first 16 hex digits of the sha1 over the sources the library was built from (a stale binary shows here) */
SR_API const char *sr_source_digest(void);
SR_API const char *sr_last_error(void);
SR_API int sr_device_count(int *count);
SR_API int sr_ctx_create(int device_id, sr_ctx **out);
SR_API int sr_ctx_create_on_stream(int device_id, void *hip_stream, sr_ctx **out);
SR_API int sr_ctx_destroy(sr_ctx *ctx);
SR_API int sr_ctx_sync(sr_ctx *ctx);

/* device memory helpers, so a host language needs no HIP binding of its own */
SR_API int sr_dev_alloc(sr_ctx *ctx, size_t bytes, void **d_ptr);
SR_API int sr_dev_free(sr_ctx *ctx, void *d_ptr);
SR_API int sr_memcpy_h2d(sr_ctx *ctx, void *d_dst, const void *h_src, size_t bytes);
SR_API int sr_memcpy_d2h(sr_ctx *ctx, void *h_dst, const void *d_src, size_t bytes);
SR_API int sr_memcpy_d2d(sr_ctx *ctx, void *d_dst, const void *d_src, size_t bytes);
SR_API int sr_memset_d(sr_ctx *ctx, void *d_dst, int value, size_t bytes);

/* per-kernel HIP-event timing on the context's stream (bench.py's roofline leg).
 * sr_prof_get: copies up to cap records; name is the kernel family, ms the summed time,
 * launches the number of launches since the last sr_prof_reset. */
typedef struct sr_prof_record {
    char name[48];
    double ms;
    int64_t launches;
} sr_prof_record;
SR_API int sr_prof_enable(sr_ctx *ctx, int on);
/* Restricts the timing to one kernel family (NULL or "": all families).  Every timed family costs two event
 * records per call, which serialise neighbouring kernels: bench.py times only the dominant kernel in its timed
 * region and all families in a separate pass. */
SR_API int sr_prof_select(sr_ctx *ctx, const char *name);
SR_API int sr_prof_reset(sr_ctx *ctx);
SR_API int sr_prof_get(sr_ctx *ctx, sr_prof_record *h_records, int cap, int *n);

/* ---- host-only bookkeeping (bit-exact integer restatements; no GPU needed) ----------- */
/* TilingModule._calculate_tile_positions (tiling_module.py:572-608).
 * h_xywh receives n_tiles x (x, y, w, h), row-major over the grid; *n_tiles is always set,
 * SR_ERR_SHAPE if cap is too small. */
SR_API int sr_tile_plan(int image_w, int image_h, int block_size, int overlap_px, int *n_tiles,
                        int *h_xywh, int cap);
/* TilingModule._calculate_overlap_for_tile (tiling_module.py:610-646): -> top,bottom,left,right */
SR_API int sr_tile_overlaps(int x, int y, int w, int h, int image_w, int image_h, int block_size,
                            int overlap_px, int *h_tblr);
/* TilingModule._build_neighbor_relationships (tiling_module.py:786-823):
 * h_nbr receives n x (top, bottom, left, right) tile indices, -1 where absent. */
SR_API int sr_tile_neighbors(const int *h_xywh, int n, int block_size, int overlap_px, int *h_nbr);
/* SuperResolutionPipeline._calculate_target_size presets (main.py:168-184); preset_mp in
 * {100,150,200}. */
SR_API int sr_target_size(int width, int height, int preset_mp, int *out_w, int *out_h);
/* BlendingModule._create_distance_weight_map (blending_module.py:529-561) tabulated by integer
 * edge distance d = 0..fw: W[y][x] = lut[min(d(y,x), fw)].  h_lut holds fw+1 floats. */
SR_API int sr_weight_lut(int fw, int weight_type, float *h_lut);

/* ---- tile extract (tiling_module.py:713-724 slice + _apply_padding :522-570) ---------- */
/* Copies n tiles out of one HWC u8 image into n contiguous block x block x cn tiles, padding
 * the bottom/right of edge tiles with the reference's border rule. */
SR_API int sr_tile_extract_pad(sr_ctx *ctx, const uint8_t *d_img, int img_h, int img_w, int cn,
                               int64_t img_stride, const int *h_xywh, int n, int block_size,
                               int pad_mode, uint8_t *d_tiles);
/* TilingModule.split_image's complexity_score = np.std(cv2.cvtColor(tile, COLOR_BGR2GRAY)) (tiling_module.py:746-749) for
 * n RGB u8 tiles of h x w that live in HBM, tile i at d_tiles + i * tile_bytes: h_sums[2 i], h_sums[2 i + 1] receive the exact
 * sums of gray and gray^2 (swap_rb != 0: the reference's BGR constants on RGB data).  No pixel leaves the GPU. */
SR_API int sr_gray_moments_u8(sr_ctx *ctx, const uint8_t *d_tiles, int n, int64_t tile_bytes, int64_t stride, int h, int w,
                              int gray_shift, int swap_rb, uint64_t *h_sums);
/* Plain overlap-tile extract (no padding): tile i = img[y:y+h, x:x+w] into d_tiles[i] with row
 * stride tile_strides[i]; used by the benchmark's output-space tile stage. */
SR_API int sr_tile_extract(sr_ctx *ctx, const uint8_t *d_img, int img_h, int img_w, int cn,
                           int64_t img_stride, const int *h_xywh, int n, void *const *h_d_tiles,
                           const int64_t *h_tile_strides);

/* ---- pyramid primitives (BlendingModule.build_gaussian_pyramid :217-269 -> cv2.pyrDown;
 *      build_laplacian_pyramid / collapse :271-363 -> cv2.pyrUp), fp32 HWC, dense ------- */
SR_API int sr_pyr_down(sr_ctx *ctx, const float *d_src, int h, int w, int cn, float *d_dst);
SR_API int sr_pyr_up(sr_ctx *ctx, const float *d_src, int hs, int ws, int cn, float *d_dst, int hd,
                     int wd);
/* d_out = d_a - pyrUp(d_b) (one Laplacian level) and d_out = pyrUp(d_b) + d_a (one collapse step) */
SR_API int sr_pyr_up_sub(sr_ctx *ctx, const float *d_a, int h, int w, int cn, const float *d_b,
                         float *d_out);
SR_API int sr_pyr_up_add(sr_ctx *ctx, const float *d_a, int h, int w, int cn, const float *d_b,
                         float *d_out);

/* ---- blends (BlendingModule.laplacian_fusion :369-506, weighted_average_fusion :661-760) */
typedef struct sr_tile_rect {
    int x, y; /* canvas position of the tile's top-left pixel (TileInfo.x, TileInfo.y) */
    int w, h; /* tile size in pixels                                                   */
} sr_tile_rect;

/* A plan owns the device workspace (per-tile Gaussian / collapsed pyramids, weight pyramids and
 * descriptor tables) for one tile arrangement.  Only canvas rows [row_begin, row_end) are
 * produced (0, canvas_h for the whole image): a strip owner in the multi-GPU blend gets
 * bit-identical rows because every pyramid value it needs is computed from the same inputs in
 * the same order. */
SR_API int sr_blend_plan_create(sr_ctx *ctx, const sr_tile_rect *h_tiles, int n, int cn, int canvas_h,
                                int canvas_w, int levels, int weight_type, int row_begin,
                                int row_end, sr_blend_plan **out);
SR_API int sr_blend_plan_destroy(sr_blend_plan *plan);
/* Host-only form of the window planner (no context, no GPU): for canvas rows [row_begin,row_end)
 * writes n x (r0, r1) = the tile-local input rows each tile must supply (r0 >= r1: tile unused).
 * The multi-GPU exchange plan is built from this on every rank. */
SR_API int sr_strip_tile_rows(const sr_tile_rect *h_tiles, int n, int levels, int canvas_h,
                              int row_begin, int row_end, int *h_rows);
/* Analytic worst case of that back-propagation (host only): a strip reads at most *below input rows before and *above
 * rows after its own rows of a tile (levels of >= 8 rows; 6 levels: 155 / 125). */
SR_API int sr_pyramid_halo(int levels, int *below, int *above);
/* Multi-GPU planning for a non-Python host (host only, deterministic, identical on every rank).
 * sr_strip_bounds: world + 1 even row boundaries of horizontal canvas strips of equal work (assessment + gather of the
 * strip's rows, pyramids of its rows plus the sr_pyramid_halo recompute).  sr_exchange_plan: the same bounds, per rank the
 * canvas rows it blends h_rows[2 r .. 2 r + 1] (strip + metric_halo rows for the SSIM windows), the tile-local rows it
 * needs of every tile h_need[(r * n + t) * 2 ..] (empty: [0, 0)) and an owner per tile.  Rank o sends rank r the rows
 * h_need[r][t] of every tile t it owns (one grouped batch of point-to-point transfers; in the order (r, t) on the sender
 * and (owner, t) on the receiver); the strip owner then calls sr_blend_plan_create(..., row_begin, row_end) with virtual
 * tile base pointers.  Policies: balanced = greedy minimisation of the busiest rank-to-rank link (xGMI is point-to-point). */
enum sr_owner_policy { SR_OWNER_BALANCED = 0, SR_OWNER_ROUNDROBIN = 1, SR_OWNER_LOCALITY = 2 };
SR_API int sr_strip_bounds(const sr_tile_rect *h_tiles, int n, int levels, int canvas_h, int canvas_w, int world,
                           int *h_bounds);
SR_API int sr_exchange_plan(const sr_tile_rect *h_tiles, int n, int cn, int levels, int canvas_h, int canvas_w, int world,
                            int metric_halo, int owner_policy, int *h_bounds, int *h_rows, int *h_need, int *h_owner);
/* The transfers (csrc/sr_comm.cpp).  SURVEY 8(b) planned sr_comm_init + a sharded blend: one process per GPU, RCCL over
 * xGMI.  No reference counterpart.  librccl.so is bound at run time (the copy the process already holds -- PyTorch's --
 * else ROCm's; SR_RCCL_LIB overrides); without it these return SR_ERR_UNSUPPORTED, RCCL failures are SR_ERR_COMM.
 *   rank 0: sr_comm_unique_id(id) -> the host ships the 128 bytes to every rank (MPI, a socket, a file) ->
 *   every rank: sr_comm_init(ctx, id, world, rank, &comm)  [collective: ncclCommInitRank on ctx's device]
 *   per image: sr_exchange_plan(...) once per geometry, then sr_comm_exchange_tile_rows(...) on ctx's stream, then
 *   sr_blend_plan_create(row_begin, row_end) + the blend + metrics on the strip, sr_comm_allreduce_f64 of the 4 partial sums.
 * sr_comm_wrap adopts an ncclComm_t the host created itself (not destroyed by sr_comm_destroy).
 * sr_comm_exchange: ONE ncclGroupStart/End around the sends and receives (u8 bytes), so pairs that send to each other
 * cannot deadlock; asynchronous on ctx's stream.  sr_comm_exchange_tile_rows builds that batch from the plan: rank `me`
 * sends rows h_need[r][t] of every tile t it owns (d_owned[t], DENSE rows of w * cn bytes) to every other rank r that needs
 * them, and receives rows h_need[me][t] of the tiles others own into d_recv[t] (a dense buffer of (r1 - r0) * w * cn bytes;
 * the blend then takes d_recv[t] - r0 * w * cn as the tile's virtual base pointer).  u8 tiles only. */
#define SR_COMM_ID_BYTES 128
typedef struct sr_comm sr_comm;
typedef struct sr_xfer {
    int peer;
    void *d_ptr;
    uint64_t bytes;
} sr_xfer;
SR_API int sr_comm_unique_id(void *id128);
SR_API int sr_comm_init(sr_ctx *ctx, const void *id128, int world, int rank, sr_comm **out);
SR_API int sr_comm_wrap(void *nccl_comm, sr_comm **out);
SR_API int sr_comm_info(const sr_comm *comm, int *world, int *rank);
SR_API int sr_comm_destroy(sr_comm *comm);
SR_API int sr_comm_exchange(sr_ctx *ctx, sr_comm *comm, const sr_xfer *sends, int n_send, const sr_xfer *recvs, int n_recv);
SR_API int sr_comm_exchange_tile_rows(sr_ctx *ctx, sr_comm *comm, const sr_tile_rect *h_tiles, int n, int cn,
                                      const int *h_need, const int *h_owner, const void *const *d_owned,
                                      const int64_t *owned_stride, void *const *d_recv);
/* SURVEY 8(b)'s sr_laplacian_blend_sharded: sr_comm_exchange_tile_rows followed, on the same stream, by sr_laplacian_blend of
 * this rank's rows -- plan = sr_blend_plan_create(..., rows[2 rank], rows[2 rank + 1]) of the exchange plan, the received
 * windows addressed through virtual base pointers.  (A host that wants the exchange of image i + 1 under the blend of image
 * i issues the two calls itself on two streams, as device_pipeline.py does.)  `plan` must have been created on `ctx` (one
 * stream orders the receive before the blend) for the same n tiles and cn channels: SR_ERR_INVALID_ARG / SR_ERR_SHAPE
 * otherwise; the tile arguments are checked by sr_sharded_tile_bases before anything is posted.  If a post inside the RCCL
 * group fails (SR_ERR_COMM) the batch is partial and the communicator must be destroyed: further calls on it fail. */
SR_API int sr_laplacian_blend_sharded(sr_ctx *ctx, sr_comm *comm, sr_blend_plan *plan, const sr_tile_rect *h_tiles, int n, int cn,
                                      const int *h_need, const int *h_owner, const void *const *d_owned,
                                      const int64_t *strides, void *const *d_recv, uint8_t *d_canvas, int64_t canvas_stride);
/* host only: the batch sr_comm_exchange_tile_rows posts, for hosts that move the rows themselves (MPI, hipMemcpyPeer).
 * Sends in (reader, tile) order, receives in (owner, tile) order; *n_send / *n_recv are the counts needed (SR_ERR_SHAPE when
 * the arrays are too small: at most n * (world - 1) sends and n receives). */
SR_API int sr_exchange_xfers(const sr_tile_rect *h_tiles, int n, int cn, int world, int rank, const int *h_need,
                             const int *h_owner, const void *const *d_owned, const int64_t *owned_stride,
                             void *const *d_recv, sr_xfer *sends, int cap_send, int *n_send, sr_xfer *recvs, int cap_recv,
                             int *n_recv);
/* host only: the per-tile base pointers rank `rank` hands to its strip blend -- an owned tile as it is, a received row window
 * moved up to its (virtual) row 0 (d_recv[t] - r0 * w * cn).  Validates what sr_laplacian_blend_sharded relies on: owners
 * inside the world, rows inside the tile, a dense stride (w * cn) for every tile that is received (SR_ERR_SHAPE otherwise),
 * a buffer for every window.  h_base[n] receives the pointers (NULL for tiles this strip does not read). */
SR_API int sr_sharded_tile_bases(const sr_tile_rect *h_tiles, int n, int cn, int world, int rank, const int *h_need,
                                 const int *h_owner, const void *const *d_owned, const int64_t *strides,
                                 void *const *d_recv, void **h_base);
SR_API int sr_comm_allreduce_f64(sr_ctx *ctx, sr_comm *comm, double *d_buf, int count);
/* tile-local rows [*r0, *r1) of tile t that the plan reads (empty if r0 >= r1): what a strip
 * owner must hold / receive for that tile. */
SR_API int sr_blend_plan_tile_rows(const sr_blend_plan *plan, int t, int *r0, int *r1);
SR_API int sr_blend_plan_workspace_bytes(const sr_blend_plan *plan, size_t *bytes);

/* h_d_tiles[i]: device address of row 0 of tile i (rows outside sr_blend_plan_tile_rows are
 * never touched, so the address may be virtual); h_strides[i]: row stride in bytes.
 * d_canvas: u8 HWC canvas (row stride canvas_stride bytes), only rows [row_begin,row_end)
 * are written.  d_canvas_f32 (nullable): dense fp32 HWC canvas receiving the normalised value
 * before clip / truncation (parity tests). */
SR_API int sr_laplacian_blend(sr_blend_plan *plan, int dtype, void *const *h_d_tiles,
                              const int64_t *h_strides, uint8_t *d_canvas, int64_t canvas_stride,
                              float *d_canvas_f32);
/* The Laplacian blend in two stages, so a strip owner can overlap the row exchange with compute:
 * sr_blend_pyramids builds G_i / R_i for the listed tiles only (first != 0: also the weight pyramids, which depend
 * only on the plan and are built once and kept; call it first with the tiles already resident, then with the tiles
 * that arrived), sr_blend_gather is the canvas gather
 * over all tiles.  sr_laplacian_blend == pyramids(all, first) + gather. */
SR_API int sr_blend_pyramids(sr_blend_plan *plan, int dtype, void *const *h_d_tiles, const int64_t *h_strides,
                             const int *h_tile_idx, int n_idx, int first);
SR_API int sr_blend_gather(sr_blend_plan *plan, int dtype, void *const *h_d_tiles, const int64_t *h_strides,
                           uint8_t *d_canvas, int64_t canvas_stride, float *d_canvas_f32);
SR_API int sr_weighted_blend(sr_blend_plan *plan, int dtype, void *const *h_d_tiles,
                             const int64_t *h_strides, uint8_t *d_canvas, int64_t canvas_stride,
                             float *d_canvas_f32);

/* weighted_average_fusion with caller-supplied weight maps (blending_module.py:729-751): h_d_weights[i] is an
 * h_i x w_i fp32 map in HBM (row stride in bytes).  Same accumulate / normalise / clip / truncate. */
SR_API int sr_weighted_blend_custom(sr_blend_plan *plan, int dtype, void *const *h_d_tiles,
                                    const int64_t *h_strides, const float *const *h_d_weights,
                                    const int64_t *h_weight_strides, uint8_t *d_canvas, int64_t canvas_stride,
                                    float *d_canvas_f32);

/* Host-buffer conveniences with the reference's call shape (ndarrays in, ndarray out):
 * stage tiles to HBM, blend, copy the canvas back.  h_tiles[i] is a dense HWC array. */
SR_API int sr_laplacian_fusion_host(sr_ctx *ctx, int dtype, const void *const *h_tiles,
                                    const sr_tile_rect *h_rects, int n, int cn, int canvas_h,
                                    int canvas_w, int levels, int weight_type, uint8_t *h_canvas,
                                    float *h_canvas_f32);
SR_API int sr_weighted_fusion_host(sr_ctx *ctx, int dtype, const void *const *h_tiles,
                                   const sr_tile_rect *h_rects, int n, int cn, int canvas_h,
                                   int canvas_w, int weight_type, uint8_t *h_canvas,
                                   float *h_canvas_f32);

/* ---- BlendingModule.detect_seams window scan (blending_module.py:765-903) -------------
 * For every tile, windows of `window` x `window` pixels every `stride` pixels over the part of the tile inside the
 * canvas; global-statistics SSIM (fp64) between the tile and the canvas window on gray values (the reference's
 * BGR2GRAY-on-RGB quirk included).  Windows scoring below `threshold` are returned unordered (up to cap records;
 * *h_count is the number found -- call again with a larger buffer if it exceeds cap).  Merging adjacent windows into
 * Seam boxes is host bookkeeping (blending_module.py:905-967). */
typedef struct sr_seam_record {
    int tile, x, y, pad;     /* tile index, canvas position of the window */
    double score;
} sr_seam_record;
SR_API int sr_seam_scan(sr_ctx *ctx, const uint8_t *d_canvas, int64_t canvas_stride, int canvas_h, int canvas_w,
                        int cn, const sr_tile_rect *h_rects, void *const *h_d_tiles, const int64_t *h_strides,
                        int n, int window, int stride, int gray_shift, double threshold, sr_seam_record *h_out,
                        int cap, int *h_count);

/* ---- TilingModule.merge_tiles feather path (tiling_module.py:1074-1175) ---------------- */
typedef struct sr_merge_tile {
    int x, y;                 /* int(global_x*scale), int(global_y*scale)                 */
    int src_w, src_h;         /* size of the tile data as handed in                       */
    int out_w, out_h;         /* metadata.output_w/h: data is bilinearly resized to this  */
    int ov_t, ov_b, ov_l, ov_r; /* int(overlap*scale) ramps; 0 = none                     */
} sr_merge_tile;
SR_API int sr_feather_merge(sr_ctx *ctx, const sr_merge_tile *h_tiles, int n, void *const *h_d_tiles,
                            const int64_t *h_strides, int blending, uint8_t *d_canvas,
                            int64_t canvas_stride, int canvas_h, int canvas_w);
/* the same with the tiles' element type given: SR_U8 (as above) or SR_F32 -- float32 tile data is accumulated as it is
 * (the reference's astype(float32), tiling_module.py:1104-1109) or goes through cv2.resize's float INTER_LINEAR arithmetic
 * when its size differs from out_w x out_h.  Strides in bytes. */
SR_API int sr_feather_merge_dt(sr_ctx *ctx, int dtype, const sr_merge_tile *h_tiles, int n, void *const *h_d_tiles,
                               const int64_t *h_strides, int blending, uint8_t *d_canvas, int64_t canvas_stride,
                               int canvas_h, int canvas_w);

/* ---- quality metrics (quality_assessment_module.py:277-417) ---------------------------- */
/* Sum of squared differences over h rows of rowlen u8 elements -> *h_sse (exact integer).
 * PSNR = 10 log10(data_range^2 / (sse / (h*rowlen))) is finished on the host (and partial sums
 * of strips are simply added). */
SR_API int sr_sse_u8(sr_ctx *ctx, const uint8_t *d_a, int64_t stride_a, const uint8_t *d_b,
                     int64_t stride_b, int h, int64_t rowlen, uint64_t *h_sse);
/* Asynchronous form: result left in device memory (8 bytes), no stream sync. */
SR_API int sr_sse_u8_async(sr_ctx *ctx, const uint8_t *d_a, int64_t stride_a, const uint8_t *d_b,
                           int64_t stride_b, int h, int64_t rowlen, uint64_t *d_sse);
SR_API double sr_psnr_from_sse(uint64_t sse, uint64_t count, double data_range);
/* fp32 images (calculate_psnr on float arrays with max > 1, quality_assessment_module.py:191-195 leaves them float):
 * skimage semantics, fp32 difference and square, fp64 sum.  PSNR = 10 log10(data_range^2 / (sse / count)). */
SR_API int sr_sse_f32(sr_ctx *ctx, const float *d_a, int64_t stride_a, const float *d_b, int64_t stride_b,
                      int h, int64_t rowlen, double *h_sse);

/* SSIM between two u8 images of h x w pixels with cn channels (cn == 3: RGB -> gray with the
 * OpenCV fixed-point rule, gray_shift 15 or 14; cn == 1: already gray).  The SSIM map is
 * summed over map rows [row_begin,row_end) intersected with the mode's valid region; the sum
 * and the number of map samples come back so strips can be added.  mean = sum / count. */
SR_API int sr_ssim_u8(sr_ctx *ctx, const uint8_t *d_a, int64_t stride_a, const uint8_t *d_b,
                      int64_t stride_b, int h, int w, int cn, int mode, int gray_shift,
                      double data_range, int row_begin, int row_end, double *h_sum,
                      uint64_t *h_count);
/* The same three SSIM variants on FLOAT images (the reference passes float arrays with max > 1 on unchanged,
 * quality_assessment_module.py:169-195,351-417): dtype SR_F32 or SR_F64, cn 1 (gray as it is) or 3 (float32 only: cv2's
 * float RGB2GRAY, 0.299 R + 0.587 G + 0.114 B in fp32; cv2.cvtColor rejects float64).  float64 arithmetic throughout, like
 * scikit-image 0.18.3 (the version pinned here); a separable reference form, not a tuned kernel.  Strides in bytes.
 * PARITY UNPINNED for SR_F32: scikit-image >= 0.19 (the reference asks for >= 0.21) keeps float32 inputs in float32
 * (_supported_float_type), so its filters and SSIM algebra run in fp32 there and differ from this float64 form by about
 * 1e-6 relative -- inside the north star's 1e-4, outside the 1e-9 the tests hold against this repository's own oracle. */
SR_API int sr_ssim_float(sr_ctx *ctx, int dtype, const void *d_a, int64_t stride_a, const void *d_b, int64_t stride_b,
                         int h, int w, int cn, int mode, double data_range, int row_begin, int row_end, double *h_sum,
                         uint64_t *h_count);
SR_API int sr_ssim_u8_async(sr_ctx *ctx, const uint8_t *d_a, int64_t stride_a, const uint8_t *d_b,
                            int64_t stride_b, int h, int w, int cn, int mode, int gray_shift,
                            double data_range, int row_begin, int row_end, double *d_sum,
                            uint64_t *h_count);
/* Fused assessment: ONE pass over both images (6 bytes / pixel) for the squared-difference sum and
 * the Gaussian SSIM in both variants (cropped "gauss11" and full-frame REFLECT_101 "simple" share all
 * filtered values), plus an all-integer pass for the uniform-7 variant.  Every field is a partial sum over
 * rows [row_begin,row_end) -- additive across strips; divide by sr_ssim_count / (h*w*cn).  The record
 * is all fp64 (the SSE is an exact integer < 2^53) so a strip owner can all-reduce it as is. */
enum sr_assess_flags { SR_ASSESS_SSE = 1, SR_ASSESS_UNIFORM7 = 2, SR_ASSESS_GAUSS11 = 4, SR_ASSESS_SIMPLE = 8, SR_ASSESS_ALL = 15 };
typedef struct sr_assess_sums {
    double sse;
    double ssim_uniform;
    double ssim_gauss;
    double ssim_simple;
} sr_assess_sums;
SR_API int sr_assess_u8_async(sr_ctx *ctx, const uint8_t *d_a, int64_t stride_a, const uint8_t *d_b,
                              int64_t stride_b, int h, int w, int cn, int gray_shift, double data_range,
                              int row_begin, int row_end, int flags, sr_assess_sums *d_out);
SR_API int sr_assess_u8(sr_ctx *ctx, const uint8_t *d_a, int64_t stride_a, const uint8_t *d_b,
                        int64_t stride_b, int h, int w, int cn, int gray_shift, double data_range,
                        int row_begin, int row_end, int flags, sr_assess_sums *h_out);
/* The same sums taken on the cv2.INTER_CUBIC resize of both images to dst_h x dst_w, sampled on the fly -- the
 * resized images are never written to memory.  One call per scale replaces downsample_bicubic x 2 + PSNR + SSIM of
 * QualityAssessmentModule._evaluate_downsample_comparison (quality_assessment_module.py:518-555, 226-253).
 * a, b: h x w x cn u8; divide by sr_ssim_count(dst_h, dst_w, ...) / (dst_h * dst_w * cn). */
SR_API int sr_assess_resized_u8_async(sr_ctx *ctx, const uint8_t *d_a, int64_t stride_a, const uint8_t *d_b,
                                      int64_t stride_b, int h, int w, int cn, int dst_h, int dst_w,
                                      int gray_shift, double data_range, int flags, sr_assess_sums *d_out);
SR_API int sr_assess_resized_u8(sr_ctx *ctx, const uint8_t *d_a, int64_t stride_a, const uint8_t *d_b,
                                int64_t stride_b, int h, int w, int cn, int dst_h, int dst_w, int gray_shift,
                                double data_range, int flags, sr_assess_sums *h_out);
/* number of SSIM-map samples of `mode` inside rows [row_begin,row_end) (host only) */
SR_API int sr_ssim_count(int h, int w, int mode, int row_begin, int row_end, uint64_t *count);
/* cv2.cvtColor(RGB2GRAY) on u8 (quality_assessment_module.py:359-360) */
SR_API int sr_rgb2gray_u8(sr_ctx *ctx, const uint8_t *d_rgb, int64_t stride, int h, int w,
                          int gray_shift, uint8_t *d_gray, int64_t gray_stride);
/* cv2.resize(..., INTER_CUBIC) on u8 (downsample_bicubic :226-253; also the benchmark's SR stub) */
SR_API int sr_resize_cubic_u8(sr_ctx *ctx, const uint8_t *d_src, int64_t src_stride, int h, int w,
                              int cn, uint8_t *d_dst, int64_t dst_stride, int dh, int dw);
/* Same sampling, but only the dst window [x0,x0+ww) x [y0,y0+wh) of the virtual dh x dw result is
 * produced (dense, stride dst_stride): the SR stub's per-tile form. */
SR_API int sr_resize_cubic_window_u8(sr_ctx *ctx, const uint8_t *d_src, int64_t src_stride, int h,
                                     int w, int cn, int dh, int dw, int x0, int y0, int ww, int wh,
                                     uint8_t *d_dst, int64_t dst_stride);

/* ---- BlendingModule.color_correction (blending_module.py:969-1146) ---------------------------------
 * sr_histogram_u8: per-channel 256-bin histogram (np.histogram(ch, 256, [0, 256]) of u8 data), h_hist = cn x 256 counts.
 * The matching itself is 256-entry bookkeeping on the host (CDFs in float64, argmin; or the mean / std affine map of
 * _mean_std_matching) and arrives here as a per-channel table h_glut[c][v] = corrected float value of source value v.
 * sr_color_correct_u8: out = astype(u8)(clip(corrected, 0, 255)) with corrected = h_glut[c][img]; with local_filter == 1
 * corrected goes through _simple_guided_filter(guide = corrected, src = img, radius, eps) first (:1110-1146: five
 * cv2.blur((radius, radius)) box means -- anchor radius / 2, REFLECT_101 -- and fp32 element-wise algebra); with
 * local_filter == 2 through the cv2.ximgproc.guidedFilter branch of :1108-1111 instead ((2 radius + 1)^2 window,
 * BORDER_REFLECT, colour guide: per-pixel 3 x 3 covariance inverse; 1 or 3 channels; restated, parity unpinned).
 * radius 1..16.  d_out may be d_img (in place).  At the reference's radius 8 with an integer-valued table both branches run as
 * fused kernels (the a / b coefficient planes of branch 1 never leave the CU); branch 1 also with a float table (mean_std)
 * whose first-stage box sums are exact in fp64 -- sr_color_table_class says which; results do not depend on which kernels run.
 * sr_color_table_class (host only): *cls = 1 when every entry of h_glut[cn][256] is a whole number 0..255 (sums slide as
 * 32-bit integers), 2 when a sum of `terms` values of each of g = T[v], fl32(g * v), fl32(g * g) is exact in fp64 whatever
 * the order (all values multiples of 2^lo and below 2^hi with hi + log2(terms) - lo <= 53: sums slide in fp64), else 0
 * (ordered sums, cv::boxFilter's order).  terms = radius^2 for branch 1. */
SR_API int sr_histogram_u8(sr_ctx *ctx, const uint8_t *d_img, int64_t stride, int h, int w, int cn, uint64_t *h_hist);
SR_API int sr_color_table_class(const float *h_glut, int cn, int terms, int *cls);
SR_API int sr_color_correct_u8(sr_ctx *ctx, const uint8_t *d_img, int64_t stride, int h, int w, int cn,
                               const float *h_glut, int local_filter, int radius, float eps, uint8_t *d_out,
                               int64_t out_stride);

/* ---- output writers: stage 5 of SuperResolutionPipeline.process (main.py:399-404) -----------------------
 * Host-only (no context, no GPU), multi-threaded (threads <= 0: every hardware thread).  h_img: h x w x cn u8, row stride in
 * bytes.  TIFF: LZW-compressed strips (Pillow compression='tiff_lzw'; BigTIFF above 4 GB); PNG: deflate level `level`
 * (Pillow compress_level=3), filter None; JPEG: baseline YCbCr 4:2:0 (gray for cn 1) with libjpeg's quality scaling,
 * colour conversion, down-sampling, islow DCT and standard Huffman tables (Pillow quality=95 defaults), restart marker per
 * MCU row.  The lossless files decode to the input bytes; the JPEG decodes to what Pillow's own file decodes to. */
SR_API int sr_encode_tiff_lzw(const uint8_t *h_img, int h, int w, int cn, int64_t stride, const char *path, int threads);
SR_API int sr_encode_png(const uint8_t *h_img, int h, int w, int cn, int64_t stride, int level, const char *path, int threads);
SR_API int sr_encode_jpeg(const uint8_t *h_img, int h, int w, int cn, int64_t stride, int quality, const char *path,
                          int threads);

/* ---- LPIPS (quality_assessment_module.py:419-465 calculate_lpips, :197-224 _to_lpips_tensor, :135-146 model init) -------
 * The reference delegates to the package `lpips` (requirements.txt:16): net 'alex' or 'vgg', version 0.1.  Weights are
 * supplied by the caller (nothing is fetched): h_conv_w[i] = i-th convolution of the backbone, dense OIHW fp32,
 * h_conv_b[i] its bias; h_lin_w[k] = the C_k weights of the k-th 1x1 `lin` layer; h_shift / h_scale = ScalingLayer
 * constants (NULL: the package's).  AlexNet has 5 convolutions, VGG16 13.
 * sr_lpips_u8: a, b are h x w x cn u8 images in HBM (cn 1: gray repeated, 4: alpha dropped).  The image is streamed in
 * `tile` x `tile` input tiles (multiple of 16; <= 0: one tile) so a 200 MP image fits; tiles [tile_begin, tile_end) of the
 * row-major tile grid are processed (tile_end < 0: all) -- ranks of a multi-GPU job take disjoint tile ranges and add the
 * sums.  h_layer_sums[k] receives the sum of the k-th tap's lin map over those tiles; LPIPS = sum_k sums[k] / (H_k W_k)
 * with the map sizes from sr_lpips_layer_sizes (h_hw: 5 x (H, W)). */
enum sr_lpips_net { SR_LPIPS_ALEX = 0, SR_LPIPS_VGG = 1 };
typedef struct sr_lpips_model sr_lpips_model;
SR_API int sr_lpips_create(sr_ctx *ctx, int net, const float *const *h_conv_w, const float *const *h_conv_b, int n_conv,
                           const float *const *h_lin_w, int n_lin, const float *h_shift, const float *h_scale,
                           sr_lpips_model **out);
SR_API int sr_lpips_destroy(sr_lpips_model *model);
SR_API int sr_lpips_layer_sizes(int net, int h, int w, int *h_hw);
SR_API int sr_lpips_tile_count(int h, int w, int tile, int *n_tiles);
SR_API int sr_lpips_u8(sr_lpips_model *model, const uint8_t *d_a, int64_t stride_a, const uint8_t *d_b, int64_t stride_b,
                       int h, int w, int cn, int tile, int tile_begin, int tile_end, double *h_layer_sums);

#ifdef __cplusplus
}
#endif
#endif /* SR_HIP_H */
