/*
 * covahip.h -- C-ABI of libcovahip.so, the MI355X (gfx950) compressed-domain filter
 * stage for CoVA.
 *
 * This is the drop-in boundary: every entry point replaces one piece of arithmetic
 * or host state that a CoVA GStreamer element performs today, and is what that
 * element's FFI (Rust `extern "C"` / C++ direct call) would bind.  The shape follows
 * the reference's own C-ABI precedent, cova-rs/nvdsbbox/nvdsbbox.h:7-14 (opaque
 * handle, plain scalars, caller-owned byte buffers, integer status).
 *
 *   - no C++/HIP/torch types in any signature: plain pointers, sizes, ints
 *   - every function returns a covahip_status (0 = OK) unless noted
 *   - no exceptions or unwinding cross the boundary, no global state
 *   - one covahip_ctx per GPU per thread of use; host objects (stack / sort /
 *     gopfilter) are one per stream, like the element instances they back
 *   - pointers tagged "dev" must be device memory of the ctx's GPU (from
 *     covahip_malloc or any HIP allocation of the same process); "host" pointers
 *     are ordinary memory.  mem_kind says which one a dual-use pointer is.
 *
 * Reference paths below are relative to /root/reference.
 */
#ifndef COVAHIP_H
#define COVAHIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------ status */
typedef enum covahip_status {
    COVAHIP_OK = 0,
    COVAHIP_ERR_INVALID_ARG = 1,
    COVAHIP_ERR_NO_DEVICE = 2,      /* no HIP device / HIP runtime failure at init      */
    COVAHIP_ERR_HIP = 3,            /* a HIP call failed; see covahip_last_hip_error    */
    COVAHIP_ERR_NOT_LOADED = 4,     /* BlobNet weights not loaded                       */
    COVAHIP_ERR_UNSUPPORTED = 5,    /* geometry outside what the kernels are built for  */
    COVAHIP_ERR_BAD_WEIGHTS = 6,    /* weight blob header / size mismatch               */
    COVAHIP_ERR_OVERFLOW = 7,       /* caller buffer too small                          */
    COVAHIP_ERR_BAD_DATA = 8        /* malformed bincode input                          */
} covahip_status;

const char *covahip_strerror(int status);
/* "covahip <version> gfx950 ..." -- static string */
const char *covahip_version(void);

enum { COVAHIP_MEM_HOST = 0, COVAHIP_MEM_DEVICE = 1 };

/* ------------------------------------------------------------ GPU context */
typedef struct covahip_ctx covahip_ctx;

int covahip_device_count(int *count);
/* PCI address ("0000:c1:00.0") of HIP device `device_id` into out (>= 13 bytes): what a host process needs to find the
 * device's NUMA node (/sys/bus/pci/devices/<address>/numa_node, local_cpulist) and pin its per-stream threads next to the GPU
 * it feeds -- one process per GPU, as the reference runs one pipeline per GoP range (gst-gopsplit/gstgopsplit.cpp:556-603). */
int covahip_device_pci_bus_id(int device_id, char *out, int out_len);
/* Creates a context on GPU `device_id` with its own HIP stream.
 * (The reference pins its engines with gpu-id, config/blobnet/amsterdam_b128.txt:6,
 *  gst-plugins/gst-maskcopy/gstmaskcopy.cpp:247.) */
int covahip_ctx_create(int device_id, covahip_ctx **out);
void covahip_ctx_destroy(covahip_ctx *ctx);
/* Blocks until everything the ctx has enqueued (all lanes, see below) is done. */
int covahip_ctx_sync(covahip_ctx *ctx);
/* Lanes = batches in flight.  The reference keeps one GPU busy with sixteen BlobNet engines, each with a batch of its own
 * (experiment/cova/config.yaml:33-34 num_mask / mask_batch_size, pipeline/cova/pipeline.py:139-181); here a ctx owns
 * n_lanes HIP streams with an activation workspace each, and consecutive covahip_filter_forward /
 * covahip_filter_forward_frames calls on DEVICE pointers (and consecutive covahip_pipe_submit calls) go to consecutive
 * lanes, so the launches of batch k+1 fill the ramps and tails of batch k's.  Rules:
 *   - such a call sees everything enqueued on the ctx before it (copies, memsets, timers);
 *   - every other entry point (covahip_ctx_sync, timers, copies, covahip_bboxcc, covahip_blobnet_forward, host-pointer
 *     calls) waits for / is ordered behind all lanes;
 *   - two device-pointer filter calls with nothing in between may run concurrently: give them separate output buffers
 *     (inputs may be shared) or call covahip_ctx_sync between them.
 * n_lanes in [1, 4]; DEFAULT 1 = strictly in call order on one stream, no hidden concurrency (the boundary the reference's
 * own FFI has, cova-rs/nvdsbbox/nvdsbbox.h:7-14).  A caller that owns one set of output buffers per batch in flight opts in
 * with covahip_ctx_set_lanes(ctx, n), n = 2 or 3: the blobnetfilter element, tools/pipe_bench and bench.py do (covahip_pipe_* slots own
 * their buffers).  Workspace per lane at 68x120, max_batch 256: about 150 MB.  Drains the ctx first. */
int covahip_ctx_set_lanes(covahip_ctx *ctx, int n_lanes);
int covahip_ctx_get_lanes(covahip_ctx *ctx, int *n_lanes);
/* Text of the last failing HIP call on this ctx ("" if none). */
const char *covahip_last_hip_error(covahip_ctx *ctx);
/* Device properties the bench reports: name (<=255 chars), CU count, HBM bytes. */
int covahip_device_info(covahip_ctx *ctx, char *name, size_t name_cap, int *num_cu, size_t *hbm_bytes);

/* Device memory helpers so a host language needs no HIP binding of its own. */
int covahip_malloc(covahip_ctx *ctx, size_t bytes, void **dev_ptr);
int covahip_free(covahip_ctx *ctx, void *dev_ptr);
int covahip_memcpy_h2d(covahip_ctx *ctx, void *dev_dst, const void *host_src, size_t bytes);
int covahip_memcpy_d2h(covahip_ctx *ctx, void *host_dst, const void *dev_src, size_t bytes);
int covahip_memset(covahip_ctx *ctx, void *dev_ptr, int value, size_t bytes);

/* HIP-event timing on the ctx's stream (bench.py's timed region and per-kernel
 * roofline numbers).  slot in [0, 16). */
int covahip_timer_start(covahip_ctx *ctx, int slot);
int covahip_timer_stop(covahip_ctx *ctx, int slot);
/* Synchronises on the stop event and returns elapsed milliseconds. */
int covahip_timer_elapsed_ms(covahip_ctx *ctx, int slot, float *ms);

/* Per-kernel profiling: when enabled every kernel launch of blobnet/bboxcc calls is
 * bracketed by HIP events on the ctx stream; covahip_profile_read then reports the
 * accumulated time and launch count per kernel name. */
int covahip_profile_enable(covahip_ctx *ctx, int on);
/* Restricts the event bracketing to one kernel name (NULL or "" = all kernels), so a
 * timed region can carry the two events of its dominant kernel only. */
int covahip_profile_filter(covahip_ctx *ctx, const char *kernel_name);
int covahip_profile_reset(covahip_ctx *ctx);
/* Fills up to cap entries; *n gets the number of distinct kernels seen. */
typedef struct covahip_kernel_time {
    char name[48];
    double total_ms;
    int64_t launches;
} covahip_kernel_time;
int covahip_profile_read(covahip_ctx *ctx, covahip_kernel_time *out, int cap, int *n);

/* ------------------------------------------------------------------ BlobNet
 * Replaces the nvinfer/TensorRT BlobNet engine and its pre/post-processing:
 *   config/blobnet/amsterdam_b128.txt:1-28 (engine, net-scale-factor 1, RGB planar,
 *   segmentation threshold 0.5), model/tasks.py:34-55 (fp16 engine, explicit batch),
 *   utils/model/{blobnet,encoder,decoder,pointwise,preprocessing}.py (the graph), gst-plugins/gst-maskcopy/gstmaskcopy.cpp:226-230
 *   (class_map + 1 -> GRAY8 {0,1} mask).
 *
 * weights: blob in the format of cova_amd/weights.py (64-byte header + fp32 payload).
 * h_mb x w_mb: macroblock grid (e.g. 68x120 for 1080p, 45x80 for 720p); t must be 4.
 * max_batch sizes the activation workspace held in HBM by the ctx.
 * Limits of the kernels, all checked HERE (COVAHIP_ERR_UNSUPPORTED), never at forward time:
 *   16 <= h_mb, w_mb <= 1024; w_mb a multiple of 4 (the first level reads 16-byte groups of four
 *   macroblocks); one band of every level must fit the 160 KB of LDS of a CU (holds far beyond 4K grids).
 * A ctx holds ONE model: loading again replaces it.  Use one ctx per model (element instance).
 * A failed load leaves the ctx without a model (later calls return COVAHIP_ERR_NOT_LOADED).  */
int covahip_blobnet_load(covahip_ctx *ctx, const void *weights, size_t weights_bytes, int h_mb, int w_mb,
                         int t, int max_batch);
/* rgba_stack: u8 [batch][t*h_mb][w_mb][4] -- metapreprocess output (row block k =
 *   frame i-k; byte 0/1/2 = mb_type/mv_x/mv_y, byte 3 ignored).
 * logits (may be NULL): f32 [batch][h_mb][w_mb] pre-sigmoid output.
 * mask   (may be NULL): u8  [batch][h_mb][w_mb], 1 where sigmoid(logit) > 0.5.
 * mem_kind applies to all three pointers.  Asynchronous for device pointers (use
 * covahip_ctx_sync); synchronous for host pointers.                               */
int covahip_blobnet_forward(covahip_ctx *ctx, const uint8_t *rgba_stack, int batch, float *logits,
                            uint8_t *mask, int mem_kind);
/* Algorithmic MACs per frame of the loaded geometry. */
int covahip_blobnet_macs_per_frame(covahip_ctx *ctx, int64_t *macs);
/* ------------------------------------------------------------------- bboxcc
 * Replaces regionprops() (cova-rs/gst-plugins/src/bboxcc/process.rs:5-49): 8-connected
 * components with stats on an h x w u8 mask (non-zero = foreground), components in
 * OpenCV label order, keep pixel-count >= area_thresh.                            */
typedef struct covahip_box {
    int32_t left, top, width, height; /* CC_STAT_LEFT/TOP/WIDTH/HEIGHT */
    int32_t area_px;                  /* CC_STAT_AREA (pixel count)     */
} covahip_box;

/* mask: u8 [batch][h][w]; boxes: [batch][max_boxes]; counts: i32 [batch] = number of
 * components that pass the filter (if > max_boxes only the first max_boxes are
 * written).  mem_kind applies to mask, boxes and counts.
 * Limit (COVAHIP_ERR_UNSUPPORTED): w <= 256.  Frames of up to about 6,400 2x2 blocks (1080p = 68x120, 1440p =
 * 90x160 macroblock grids) keep their union-find in the LDS of one CU; larger ones (a 4K grid is 135x240) run the
 * same algorithm with that state in global memory -- same results, slower.        */
int covahip_bboxcc(covahip_ctx *ctx, const uint8_t *mask, int batch, int h, int w, int area_thresh,
                   covahip_box *boxes, int32_t *counts, int max_boxes, int mem_kind);

/* Fused hot path = nvinfer(BlobNet) -> maskcopy -> bboxcc for one batch: the mask
 * stays on the GPU.  logits/mask may be NULL.  bboxcc's limit applies on top of the model's: a grid wider than 256
 * macroblocks loads (covahip_blobnet_forward works) but this call returns COVAHIP_ERR_UNSUPPORTED.   */
int covahip_filter_forward(covahip_ctx *ctx, const uint8_t *rgba_stack, int batch, int area_thresh,
                           covahip_box *boxes, int32_t *counts, int max_boxes, float *logits,
                           uint8_t *mask, int mem_kind);

/* The same hot path fed with CARRIER frames instead of stacks: metapreprocess' temporal stacking (timestep 4,
 * cova-rs/gst-plugins/src/metapreprocess/imp.rs:288-332) becomes an index gather on the GPU.  With gamma = 1 a
 * carrier frame is a slice of four consecutive stacks; here it crosses PCIe / HBM once and the first encoder
 * level's convolution runs once per carrier frame instead of once per (stack, slice).  Results are bit-identical
 * to covahip_filter_forward on the stacks those indices describe.
 *   frames:      u8 [n_frames][h_mb][w_mb][4], any mix of streams (mem_kind as for the other pointers)
 *   stack_index: HOST i32 [batch][4]: for output b the indices into `frames` of its T = 0 (current), 1, 2, 3
 *                (oldest) slices; NULL = one stream in order (batch == n_frames - 3, output b = frames b+3 .. b)
 *   4 <= n_frames <= 4 * max_batch; an index outside [0, n_frames) is COVAHIP_ERR_INVALID_ARG.               */
int covahip_filter_forward_frames(covahip_ctx *ctx, const uint8_t *frames, int n_frames, const int32_t *stack_index,
                                  int batch, int area_thresh, covahip_box *boxes, int32_t *counts, int max_boxes,
                                  float *logits, uint8_t *mask, int mem_kind);

/* Pipelined host-buffer form of the carrier-frame hot path, for a caller that batches frames continuously (the
 * batching element gst/gstcova.c `blobnetfilter`; stands where nvstreammux -> nvinfer -> nvstreamdemux -> maskcopy
 * -> bboxcc stand in pipeline/cova/pipeline.py:139-261).  A pipe owns n_slots batches in flight: H2D of batch k+1,
 * the kernels of batch k and D2H of batch k-1 overlap.  (Round 6: the runtime multiplexes its streams onto a few hardware queues and a
 * copy stream that shares one with a lane stalls that lane; covahip_pipe_create therefore measures -- a few milliseconds, the ctx
 * idle -- where candidate streams land for the lane count the ctx has AT THAT MOMENT, uploads on a queue without a lane and sends a
 * batch's results out on the lane that ran it: set the lanes before creating the pipe.)  Per batch:
 *   acquire: a free slot and its PINNED host buffers -- frames u8 [max_frames][h_mb][w_mb][4] and stack_index
 *            i32 [max_batch][4] (see covahip_filter_forward_frames) -- which the caller fills in place;
 *            COVAHIP_ERR_OVERFLOW when every slot is taken (collect one first);
 *   submit:  enqueues copy-in, kernels, on-device compaction of the boxes and copy-out; returns at once;
 *   collect: waits for that slot; counts i32 [batch] (components that pass the filter), offsets i32 [batch + 1] and
 *            boxes [offsets[batch]] = the first min(count, max_boxes) boxes of every frame, packed; mask u8
 *            [batch][h_mb][w_mb] when the pipe was created with want_mask (else NULL);
 *   release: the caller is done with the results, the slot can be acquired again.
 * acquire / submit / collect / release: one thread at a time per pipe and its ctx (the caller's lock).
 * covahip_pipe_wait only blocks until a submitted slot's results have landed in host memory; it may run on
 * another thread, concurrently with acquire / submit of other slots (collect returns at once after it).   */
typedef struct covahip_pipe covahip_pipe;
int covahip_pipe_create(covahip_ctx *ctx, int max_batch, int max_frames, int max_boxes, int n_slots, int want_mask,
                        covahip_pipe **out);
void covahip_pipe_destroy(covahip_pipe *pipe);
/* Before the first acquire: the slots' frame area holds packed records (covahip_carrier_pack), hw * 2 bytes per carrier frame. */
int covahip_pipe_set_packed(covahip_pipe *p, int on);
/* Before the first acquire: covahip_pipe_wait / covahip_pipe_collect SLEEP until a slot's results have landed instead of spinning
 * on the completion signal (the default).  Round 6: as a poll -- hipEventQuery + a 20 us nanosleep -- because the runtime's own
 * blocking wait spins before it parks and, at a batch every 110 - 200 us, never parks (12.5 % of the plugin chain's CPU).  For callers whose host cores are the scarce resource
 * (`blobnetfilter`: its collector thread waits for the GPU most of the time). */
int covahip_pipe_set_blocking_wait(covahip_pipe *p, int on);
int covahip_pipe_acquire(covahip_pipe *pipe, int *slot, uint8_t **frames, int32_t **stack_index);
int covahip_pipe_submit(covahip_pipe *pipe, int slot, int n_frames, int batch, int area_thresh);
/* Gives an ACQUIRED slot back without submitting it (after a failed covahip_pipe_submit, or when the caller shuts down with
 * a partly filled batch). */
int covahip_pipe_abort(covahip_pipe *pipe, int slot);
int covahip_pipe_wait(covahip_pipe *pipe, int slot);
int covahip_pipe_collect(covahip_pipe *pipe, int slot, const int32_t **counts, const int32_t **offsets,
                         const covahip_box **boxes, const uint8_t **mask);
int covahip_pipe_release(covahip_pipe *pipe, int slot);

/* --------------------------------------------------------- Bbox wire format
 * bincode 1.3 (default config) bytes of Vec<Bbox> / Frame as the reference's elements
 * exchange them (cova-rs/bbox/src/bbox.rs:4-14,84-90; cova-rs/bbox/src/lib.rs:8-22).  */
typedef struct covahip_bbox {
    float left, top, width, height, area; /* area = width*height (bbox.rs:23) */
    uint64_t track_id;                    /* valid iff has_track_id           */
    uint64_t timestamp;
    uint32_t class_id;
    float confidence;
    uint8_t has_track_id, has_timestamp, has_class_id, has_confidence;
} covahip_bbox;

/* Bbox::new for each CC box (process.rs:47, bbox.rs:17-29). */
void covahip_boxes_to_bbox(const covahip_box *in, int n, covahip_bbox *out);
/* Returns the encoded size; writes only if it fits in cap (else COVAHIP_ERR_OVERFLOW
 * is reported through *status, which may be NULL). */
size_t covahip_bbox_serialize_vec(const covahip_bbox *boxes, size_t n, uint8_t *out, size_t cap, int *status);
/* Decodes up to cap boxes; *n gets the vector length found in the stream. */
int covahip_bbox_deserialize_vec(const uint8_t *data, size_t len, covahip_bbox *out, size_t cap, size_t *n);
size_t covahip_frame_serialize(uint64_t range_start, uint64_t oldest, const covahip_bbox *boxes, size_t n,
                               uint8_t *out, size_t cap, int *status);
/* Bbox::iou (bbox.rs:39-56). */
float covahip_bbox_iou(const covahip_bbox *a, const covahip_bbox *b);

/* ------------------------------------------------------ sink formats, track export
 * Data formats either side of the hot path.               */
/* tfrecordsink (cova-rs/gst-plugins/src/tfrecordsink/imp.rs:69-198): one framed TFRecord record =
 * tf.train.Example with bytes_list features mb_type / mv_x / mv_y / gt, one w*h string per frame,
 * zero-filled to pad_to_frames strings (the `gop` property).  rgba: [n_frames][h][w][4] (a
 * metapreprocess timestep=1 frame), gt: [n_frames][h*w] or NULL.  Returns the record size. */
size_t covahip_tfrecord_example(const uint8_t *rgba, const uint8_t *gt, int n_frames, int pad_to_frames, int w, int h,
                                uint8_t *out, size_t cap, int *status);
/* bboxsink (cova-rs/gst-plugins/src/bboxsink/imp.rs:252-270): serde-CSV text, optional header. */
size_t covahip_bbox_csv(const covahip_bbox *boxes, size_t n, int with_header, char *out, size_t cap, int *status);
/* cova track export (cova-rs/gst-plugins/src/cova/tracker.rs:59-83): per dead track a 4-byte
 * big-endian length + bincode Frame{range_start, oldest, history}; boxes/track_lens as returned by
 * covahip_sort_update / covahip_sort_finalize. */
size_t covahip_tracks_export(uint64_t range_start, uint64_t oldest, const covahip_bbox *boxes, const uint32_t *track_lens,
                             size_t n_tracks, uint8_t *out, size_t cap, int *status);

/* ------------------------------------------------ analysis-aggregator join
 * Association of tracker output with DNN detections (cova-rs/analysis-aggregator/src/server/
 * assoc.rs:63-507, track.rs:47-66, dnn.rs:57-86): what the aggregator
 * does with the messages of its tracker and DNN connections, without the sockets.  Messages are
 * pushed in arrival order; the four CSV files (track, dnn, assoc, stationary) are read back as text. */
typedef struct covahip_assoc covahip_assoc;
typedef struct covahip_assoc_cfg {   /* main.rs:32-39 */
    float moving_iou;                /* 0.15 */
    float stationary_iou;            /* 0.3  */
    uint64_t stationary_maxage_s;    /* 120  */
    float scale_factor;              /* 1.3  */
} covahip_assoc_cfg;
void covahip_assoc_default_cfg(covahip_assoc_cfg *cfg);
/* range_starts: the range_start every tracker announces with its first frame (assoc.rs:473-489). */
int covahip_assoc_new(const covahip_assoc_cfg *cfg, const uint64_t *range_starts, size_t n_trackers, covahip_assoc **out);
void covahip_assoc_free(covahip_assoc *a);
/* Recieved::Track: boxes already in pixels with re-based ids (assoc.rs:370-431). */
int covahip_assoc_push_track(covahip_assoc *a, uint64_t range_start, uint64_t oldest, const covahip_bbox *boxes, size_t n);
/* One length-delimited payload of covahip_tracks_export as track.rs:47-66 handles it: bincode Frame,
 * scale_dim(16), track_id += range_start, then push_track. */
int covahip_assoc_push_track_frame(covahip_assoc *a, const uint8_t *payload, size_t len);
/* Recieved::Dnn (assoc.rs:296-367); boxes need timestamp and class_id. */
int covahip_assoc_push_dnn(covahip_assoc *a, const covahip_bbox *boxes, size_t n);
/* Detection rows "timestamp,left,top,width,height,class_id\n" as read from a DNN connection (dnn.rs:57-86). */
int covahip_assoc_push_dnn_text(covahip_assoc *a, const char *text, size_t len);
int covahip_assoc_terminate(covahip_assoc *a);   /* assoc.rs:434-467 */
/* which: 0 track.csv, 1 dnn.csv, 2 assoc.csv, 3 stationary.csv; returns the size, copies if it fits. */
size_t covahip_assoc_csv(covahip_assoc *a, int which, char *out, size_t cap, int *status);

/* --------------------------------------------- entropy-decode front end
 * What feeds `metapreprocess` in the reference is a patched FFmpeg avdec_h264 (an un-vendored submodule; README.md:94-114)
 * that stops after entropy decoding and writes one record [mb_type, mv_x, mv_y, -] per macroblock into the first bytes of its
 * output frame.  Built here, verified on the reference's demo/1m.mp4: ISO-BMFF / NAL / SPS / PPS / slice-header layer, picture
 * order (output order of the access units), the CABAC macroblock layer of frame-coded 4:2:0 streams with one slice per
 * picture and cabac_init_idc 0 (every slice must end on its last macroblock with end_of_slice_flag: 1,802 of 1,802 do) and
 * the record writer.  Anything else (CAVLC, fields / MBAFF, several slices per picture, cabac_init_idc 1 / 2, samples of more than 8 bits) returns
 * COVAHIP_ERR_UNSUPPORTED.  `file` must stay valid while the handle lives. */
typedef struct covahip_h264 covahip_h264;
typedef struct covahip_h264_info {
    int32_t width_mbs, height_mbs, n_samples;
    int32_t profile_idc, level_idc, entropy_cabac, transform_8x8, num_ref_frames, frame_mbs_only;
    int32_t weighted_pred, weighted_bipred, poc_type;
    int32_t max_num_reorder_frames, max_dec_frame_buffering;   /* VUI bitstream_restriction (E.1.1); -1 when the stream does not say */
} covahip_h264_info;
typedef struct covahip_h264_slice {
    uint64_t nal_offset;       /* file offset of the NAL unit (its header byte) */
    uint32_t nal_bytes;
    uint32_t data_bit_offset;  /* first bit of slice_data() in the unescaped RBSP behind the NAL header byte */
    int32_t nal_type;          /* 1 non-IDR, 5 IDR */
    int32_t slice_type;        /* 0 P, 1 B, 2 I (slice_type % 5) */
    int32_t first_mb, frame_num, idr, poc_lsb, qp, cabac_init_idc /* -1: none */, num_ref_l0, num_ref_l1, direct_spatial;
    int32_t nal_ref_idc, has_mmco5;  /* reference picture?  memory_management_control_operation 5 present (resets the POC)? */
} covahip_h264_slice;
int covahip_h264_open_mp4(const uint8_t *file, size_t len, covahip_h264 **out);
void covahip_h264_close(covahip_h264 *h);
int covahip_h264_get_info(const covahip_h264 *h, covahip_h264_info *info);
/* Access unit `sample` (decode order): where it sits in the file and whether the container marks it a sync sample. */
int covahip_h264_sample(const covahip_h264 *h, int sample, uint64_t *offset, uint32_t *size, int *is_sync);
/* Slice headers of the access unit (up to cap; *n = number of slice NAL units). */
int covahip_h264_sample_slices(const covahip_h264 *h, int sample, covahip_h264_slice *out, int cap, int *n);
/* Access units in OUTPUT order (ascending picture order count inside every IDR period, 8.2.1): the order in which a decoder
 * hands frames downstream, i.e. the order metapreprocess stacks them in.  samples: sample indices (decode order). */
int covahip_h264_display_order(const covahip_h264 *h, int32_t *samples, int cap, int *n);
/* Entropy-decodes access unit `sample` (decode order) into records u8 [height_mbs][width_mbs][4] (may be NULL: parse only) --
 * the first width_mbs * height_mbs * 4 bytes of the carrier frame.  Byte 0: macroblock class (0 P_Skip / B_Skip, 1 inter 16x16,
 * 2 inter 16x8 / 8x16, 3 inter 8x8, 4 B_Direct_16x16, 5 intra NxN, 6 intra 16x16, 7 I_PCM); bytes 1 / 2: |mean motion vector| of
 * the macroblock, x / y, in quarter pixels (<= 255) -- the standard's prediction (median, P_Skip, spatial direct with the
 * colZeroFlag test against RefPicList1[0], temporal direct from that picture's motion; the picture is decoded on the way when no
 * earlier call has) plus the coded difference; byte 3: 0.  What the reference's patched decoder puts into these bytes is not known here (SURVEY.md row A0:
 * unpinned); a BlobNet has to be trained on the front end it runs behind.  COVAHIP_OK only if the slice decoded exactly
 * width_mbs * height_mbs macroblocks, ended there with end_of_slice_flag and left only trailing bits. */
int covahip_h264_decode_records(const covahip_h264 *h, int sample, uint8_t *records, size_t cap);
/* The co-located picture of a B picture's direct prediction: *col_sample = the sample that is RefPicList1[0] of `sample` (list
 * initialisation 8.2.4.2.3 + modification 8.2.4.3 over the reference marking 8.2.5 of the access units before it), -1 when
 * `sample` is not a B picture or has none; *short_term = 1 when that picture is a short-term reference. */
int covahip_h264_colocated(const covahip_h264 *h, int sample, int *col_sample, int *short_term);
/* Stream form (what an element in the place of avdec_h264 uses): parameter sets from the AVCDecoderConfigurationRecord (avcC box
 * payload = codec_data of video/x-h264,stream-format=avc caps), then access units IN DECODE ORDER (length-prefixed NAL units).
 * records / cap as covahip_h264_decode_records; hdr (may be NULL) gets the slice header; *order_key (may be NULL) a key whose
 * ascending order is the output order of the pictures (IDR period << 32 | picture order count + 2^31). */
int covahip_h264_open_avcc(const uint8_t *avcc, size_t len, covahip_h264 **out);
int covahip_h264_decode_au(covahip_h264 *h, const uint8_t *au, size_t len, uint8_t *records, size_t cap, covahip_h264_slice *hdr,
                           int64_t *order_key);
/* The carrier layout: interleaves per-macroblock mb_type / mv_x / mv_y into the first width_mbs * height_mbs * 4 bytes of
 * `frame` (metapreprocess/imp.rs:233,311-312; tfrecordsink/imp.rs:105-112). */
int covahip_carrier_write_records(const uint8_t *mb_type, const uint8_t *mv_x, const uint8_t *mv_y, int width_mbs, int height_mbs,
                                  uint8_t *frame, size_t frame_bytes);

/* Packed carrier records: two bytes per macroblock, min(mb_type, 6) | min(mv_x, 6) << 3 | min(mv_y, 6) << 6 -- everything BlobNet
 * keeps of a record (its first operation is clip(x, 0, 6) on all three channels, utils/model/preprocessing.py:6-7), at half the
 * bytes: the host-to-device copy is what bounds the element path (9.1 MB per 256-frame batch over PCIe).  The pinned host
 * pipeline takes its frames in this form when asked to (covahip_pipe_set_packed); results are bit-identical. */
void covahip_carrier_pack(const uint8_t *frame, size_t n_mb, uint16_t *records);
/* covahip_filter_forward_frames on packed records [n_frames][H][W] (device pointers only). */
int covahip_filter_forward_frames_packed(covahip_ctx *ctx, const uint16_t *d_records, int n_frames, const int32_t *stack_index,
                                         int batch, int area_thresh, covahip_box *d_boxes, int32_t *d_counts, int max_boxes,
                                         float *d_logits, uint8_t *d_mask);

/* ------------------------------------------------- metapreprocess stacking
 * Host state of the `metapreprocess` element (cova-rs/gst-plugins/src/metapreprocess/
 * imp.rs:204-332): keeps the last timestep-1 inputs, emits one stacked frame every
 * gamma-th input once warm.                                                       */
typedef struct covahip_stack covahip_stack;
/* size_per_buf = out_size / timestep (imp.rs:233) = (W/16)*(H/16)*4 bytes. */
int covahip_stack_new(size_t size_per_buf, unsigned timestep, unsigned gamma, covahip_stack **out);
void covahip_stack_free(covahip_stack *s);
/* in: >= size_per_buf bytes of carrier frame; out: timestep*size_per_buf bytes.
 * *emitted = 1 if `out` was written (GST_FLOW_OK), 0 if the element would return
 * BASE_TRANSFORM_FLOW_DROPPED. */
int covahip_stack_push(covahip_stack *s, const uint8_t *in, size_t in_len, uint8_t *out, size_t out_cap,
                       int *emitted);
/* transform_caps arithmetic (imp.rs:262-268): out = (w/16, h/16*timestep). */
void covahip_stack_out_dims(int width, int height, unsigned timestep, int *out_w, int *out_h);

/* ----------------------------------------------------------------- SORT
 * Host state of the `sorttracker` element and of cova's embedded tracker
 * (cova-rs/sort/src/lib.rs:14-214, tracker/mod.rs:15-152, state.rs:9-28).            */
typedef struct covahip_sort covahip_sort;
int covahip_sort_new(uint64_t max_age, uint64_t min_hits, float iou_threshold, covahip_sort **out);
void covahip_sort_free(covahip_sort *s);
/* Sort::update(dets, pts).  Dead active tracks' histories are appended, flattened in
 * track order, to dead_boxes (cap entries; *n_dead_boxes = total produced);
 * track_lens (cap_tracks entries; *n_tracks = number of dead tracks) holds each
 * track's history length.  Any output pointer may be NULL with cap 0.           */
int covahip_sort_update(covahip_sort *s, const covahip_bbox *dets, size_t n_dets, uint64_t pts,
                        covahip_bbox *dead_boxes, size_t cap, size_t *n_dead_boxes, uint32_t *track_lens,
                        size_t cap_tracks, size_t *n_tracks);
/* Sort::finalize (lib.rs:207-213): same output convention. */
int covahip_sort_finalize(covahip_sort *s, covahip_bbox *boxes, size_t cap, size_t *n_boxes,
                          uint32_t *track_lens, size_t cap_tracks, size_t *n_tracks);
int covahip_sort_mark_seen(covahip_sort *s, uint64_t ts);          /* lib.rs:189-193 */
int covahip_sort_num_trackers(const covahip_sort *s, size_t *n);
/* Introspection for tests: tracker i's id / active flag / hit_streaks / state box. */
int covahip_sort_tracker_info(const covahip_sort *s, size_t i, uint64_t *id, int *active,
                              uint64_t *hit_streaks, uint64_t *time_since_update, covahip_bbox *state);
/* Introspection for tests (the reference's own unit tests drive the tracker this way, sort/src/lib.rs:250-274,
 * tracker/mod.rs:154-165): KalmanBoxTracker::predict(ts) on tracker i (tracker/mod.rs:104-121; *last = history.last()),
 * KalmanBoxTracker::update(Some(det) / None) (tracker/mod.rs:71-102; det == NULL is None). */
int covahip_sort_tracker_predict(covahip_sort *s, size_t i, uint64_t ts, covahip_bbox *last);
int covahip_sort_tracker_update(covahip_sort *s, size_t i, const covahip_bbox *det);
/* linear_assignment() of lib.rs:25-56 on a column-major n_rows x n_cols f32 cost
 * matrix; writes (row, col) pairs; returns their number. */
size_t covahip_linear_assignment(const float *cost_colmajor, size_t n_rows, size_t n_cols, uint32_t *pairs,
                                 size_t cap_pairs);

/* ------------------------------------------------------------- cova filter
 * Host state of the `cova` element (cova-rs/gst-plugins/src/cova/imp.rs:90-432,
 * cova/tracker.rs:16-125): GoP buffering of encoded access units, embedded SORT,
 * decode/drop decisions and the three read-only counters.                        */
typedef struct covahip_gopfilter covahip_gopfilter;
typedef struct covahip_gopfilter_cfg {
    float sort_iou;        /* "sort-iou"     default 0.1  (imp.rs:22) */
    uint32_t sort_maxage;  /* "sort-maxage"  default 30   */
    uint32_t sort_minhits; /* "sort-minhits" default 30   */
    uint32_t alpha;        /* "alpha"        struct default 0 (imp.rs:28) */
    uint32_t beta;         /* "beta"         struct default 0 */
    uint8_t infer_i;       /* "infer-i"      default false */
} covahip_gopfilter_cfg;
void covahip_gopfilter_default_cfg(covahip_gopfilter_cfg *cfg);
int covahip_gopfilter_new(const covahip_gopfilter_cfg *cfg, covahip_gopfilter **out);
void covahip_gopfilter_free(covahip_gopfilter *g);

enum {
    COVAHIP_AU_DELTA_UNIT = 1u << 0, /* in:  not a key frame (GST_BUFFER_FLAG_DELTA_UNIT) */
    COVAHIP_AU_DISCONT = 1u << 1,    /* out: set on the copy of each GoP's key frame       */
    COVAHIP_AU_DROPPABLE = 1u << 2   /* out: decode for dependency only                    */
};
typedef struct covahip_au_out {
    uint64_t id;    /* caller's handle of the access unit (e.g. GstBuffer*) */
    uint64_t pts;   /* ns */
    uint32_t flags; /* COVAHIP_AU_* */
    uint32_t list;  /* index of the BufferList this AU belongs to (push order) */
} covahip_au_out;

/* sink_enc chain (imp.rs:320-360). */
int covahip_gopfilter_push_enc(covahip_gopfilter *g, uint64_t id, uint64_t pts, uint32_t flags);
/* sink_mask chain (imp.rs:90-317): boxes of the frame at `pts`; forwarded AUs are
 * appended to out (cap entries, *n_out = number produced).                        */
int covahip_gopfilter_push_boxes(covahip_gopfilter *g, const covahip_bbox *boxes, size_t n, uint64_t pts,
                                 covahip_au_out *out, size_t cap, size_t *n_out);
/* Both-sinks-EOS flush (imp.rs:361-432). */
int covahip_gopfilter_eos(covahip_gopfilter *g, covahip_au_out *out, size_t cap, size_t *n_out);
/* Access units the filter has discarded for good since the last call (GoP leftovers of a flushed GoP, the AU
 * popped and lost at imp.rs:167-172, everything still queued at EOS): the caller releases its buffers for
 * these ids, as the reference frees a GoP's buffers when it drops it (imp.rs:268-305).  Call until *n < cap. */
int covahip_gopfilter_take_dropped(covahip_gopfilter *g, uint64_t *ids, size_t cap, size_t *n);
/* Bytes the element's tracker writes to its aggregator socket (`port`; cova/tracker.rs:59-83,91-118): one
 * 4-byte big-endian length + bincode Frame per finished track, accumulated since the last successful call
 * (tracks that died in push_boxes, Sort::finalize() at EOS).  Returns the size; copies and clears if it fits. */
size_t covahip_gopfilter_take_track_export(covahip_gopfilter *g, uint8_t *out, size_t cap, int *status);
int covahip_gopfilter_counters(const covahip_gopfilter *g, uint64_t *dropped, uint64_t *decoded_dependency,
                               uint64_t *decoded_inference);

#ifdef __cplusplus
}
#endif
#endif /* COVAHIP_H */
