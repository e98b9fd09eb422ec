/*
 * rcflow.h -- C ABI of librcflow.so: MI355X (gfx950) dense Farneback optical flow and
 * the per-pixel rip-current analysis that consumes it.
 *
 * This is the drop-in boundary for the one hot path of borgor/ripcurrents (SURVEY.md
 * section 8b).  Every entry point names the reference interface it replaces; paths are
 * relative to /root/reference/RipCurrents_main.  Plain pointers and sizes only; image
 * arguments are (pointer, byte step) pairs laid out like cv::Mat (interleaved channels).
 *
 * Conventions
 *  - returns RC_OK (0) or a negative RC_E* code; nothing throws across the ABI;
 *  - the caller owns every buffer; the context owns its device workspaces;
 *  - a context is bound to one GPU; `stream` selects one of its independent stream
 *    slots (own HIP stream, workspaces and analysis state).  Calls on one slot are
 *    ordered; distinct slots may be driven from distinct host threads;
 *  - *_dev entry points take DEVICE pointers, enqueue on the slot's HIP stream and
 *    return without waiting (rcflow_sync waits); the host-pointer forms copy in,
 *    compute, copy out and return when the result is in the caller's buffer;
 *  - there is no CPU fallback: without a usable HIP device rcflow_create fails.
 */
#ifndef RCFLOW_H
#define RCFLOW_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RCFLOW_ABI_VERSION 1

typedef struct rc_ctx rc_ctx;

enum {
    RC_OK = 0,
    RC_EINVAL = -1,   /* bad argument (what cv::Exception / CV_Assert is in the reference) */
    RC_ENOMEM = -2,   /* device or host allocation failed */
    RC_EHIP = -3,     /* a HIP runtime call failed; rcflow_last_error() has the text */
    RC_ENODEV = -4,   /* no usable gfx950 device */
    RC_ESIZE = -5,    /* frame larger than the context was created for */
    RC_ESTATE = -6,   /* call order violated (e.g. analysis before any flow) */
    RC_ECOMM = -7     /* collective layer not initialised / failed */
};

/* cv::OPTFLOW_FARNEBACK_GAUSSIAN; cv::OPTFLOW_USE_INITIAL_FLOW (4) is rejected with
 * RC_EINVAL: no reference call site uses it. */
#define RC_FARNEBACK_GAUSSIAN 256

/* ripcurrents.hpp:7-9 */
#define RC_HIST_BINS 50
#define RC_HIST_DIRECTIONS 36
#define RC_HIST_RESOLUTION 20
/* hist[50] | hist2d[36*50] | histsum | histsum2d[36]: the all-reduce payload */
#define RC_HIST_WORDS (RC_HIST_BINS + RC_HIST_DIRECTIONS * RC_HIST_BINS + 1 + RC_HIST_DIRECTIONS)

/* Parameters of cv::calcOpticalFlowFarneback, in its argument order. */
typedef struct rc_farneback_params {
    double pyr_scale;
    int levels;
    int winsize;
    int iterations;
    int poly_n;
    double poly_sigma;
    int flags;
} rc_farneback_params;

/* ------------------------------------------------------------------ lifetime */

/* Creates a context on HIP device `device` with `max_streams` stream slots, each able to
 * hold frames up to max_w x max_h. */
int rcflow_create(rc_ctx** out, int device, int max_w, int max_h, int max_streams);
void rcflow_destroy(rc_ctx* ctx);
int rcflow_abi_version(void);
const char* rcflow_last_error(void);
/* Waits for everything enqueued on the slot. */
int rcflow_sync(rc_ctx* ctx, int stream);
/* Runs the slot on a caller-owned hipStream_t (e.g. the host framework's current
 * stream; NULL is the device's null stream).  rcflow_use_own_stream restores the slot's
 * own non-blocking stream. */
int rcflow_set_hip_stream(rc_ctx* ctx, int stream, void* hip_stream);
int rcflow_use_own_stream(rc_ctx* ctx, int stream);
/* "exact" (-1 | 0 | 1, default -1): 1 runs every Farneback stage in the operation order of OpenCV's CPU path
 * (float / double exactly where optflow.cpp has them, no fused multiply-adds): the flow field is then
 * bit-identical to the CPU path (tests/test_gpu_exact.py), at 2-15x the time.  0 always takes the fast kernels
 * (reordered fp32 sums: within SURVEY 8(d)'s tolerance wherever the 2x2 system is well conditioned).  -1 picks
 * exact only where the fast kernels cannot hold that tolerance: the near-pointwise windows (Gaussian winsize < 7,
 * i.e. main.cpp:264, :742, ripcurrents_module.cpp:712, main_old.cpp:324, and winsize 1), whose determinant
 * vanishes on smooth regions so that any rounding difference is amplified from scale to scale.
 * Tunables: "chunk" = frame pairs per launch in clip mode (default 32);
 * "exact_taps" = 1 keeps every polynomial-expansion tap instead of dropping taps whose
 * total weight is below 1e-8 of the kernel mass (default 0);
 * "fuse_iters" = 0 runs every Farneback iteration as its own launch instead of two per
 * launch (default 1; results are bit-identical either way);
 * "chain" = consecutive frame pairs a block of the fused winsize-3 flow kernel walks on its tile, taking pair
 * z + 1's previous-frame coefficients out of pair z's next-frame window in LDS (the reference's u_f1.copyTo(u_f2),
 * ripcurrents.cpp:194-221, at tile level; default 8, 1 = off; halved for small launches until "chain_min_blocks" blocks
 * (default 4096) remain; same bits).
 * Measurement switches, all speed-only except where noted: "xcd_remap" (1) XCD-aware tile order;
 * "poly_tile_h" (32 | 48) rows per expansion block -- changes the per-tile DC and with it the last
 * bits of R; "poly_mfma" (0) vertical pass of the expansion on the matrix cores -- different
 * summation order, same tolerance; "overlap" (0) clip path on two streams; "hist_blocks" (0 = default)
 * cap on histogram blocks; "frame_overlap" (0 | 1 | 2, default 1): the frame loop on two streams -- upload and expansion of
 * frame t+1 beside the flow kernels of frame t; 1 = rcflow_push_frame_u8 only (the library owns the upload), 2 = also
 * rcflow_push_frame_dev, the caller then guaranteeing that d_frame is complete when the call is made; "merge_small" (1) merged launches for calls of one or two frames; "fuse_pyr" (1)
 * pyramid scales 1 and 2 written by the scale-0 expansion launch (pyr_scale 0.5, exact half / quarter sizes);
 * "ablate" bit field selecting alternative kernel forms for A/B runs and the bit-identity tests (2048 byte-wise u8
 * staging, 4096 earlier pyramid kernels, 8192 generic large-window kernel, 65536 tile kernel for Gaussian winsize
 * 10 / 20, 33554432 tile chains whatever the launch size; the full list is enum RcAblate in csrc/rc_common.h);
 * "stamps" diagnostic builds only. */
int rcflow_set_option(rc_ctx* ctx, const char* name, int value);

/* ------------------------------------------------------------------ A: Farneback
 * Replaces  cv::calcOpticalFlowFarneback(prev, next, flow, pyr_scale, levels, winsize,
 *           iterations, poly_n, poly_sigma, flags)
 * as called at ripcurrents.cpp:215, main.cpp:264,609,742,961,1119,1481,
 * ripcurrents_module.cpp:712, main_old.cpp:324 (8UC1 in, CV_32FC2 out). */
int rcflow_farneback_u8(rc_ctx* ctx, int stream, const uint8_t* prev, size_t prev_step,
                        const uint8_t* next, size_t next_step, int w, int h, float* flow_xy,
                        size_t flow_step, double pyr_scale, int levels, int winsize,
                        int iterations, int poly_n, double poly_sigma, int flags);
/* Same, device pointers, asynchronous. */
int rcflow_farneback_dev(rc_ctx* ctx, int stream, const uint8_t* d_prev, size_t prev_step,
                         const uint8_t* d_next, size_t next_step, int w, int h,
                         float* d_flow_xy, size_t flow_step, const rc_farneback_params* p);
/* Streaming form of the frame loop ripcurrents.cpp:194-221 (`u_f1.copyTo(u_f2)` at :216):
 * the slot keeps the previous frame's polynomial expansion, so each call does one
 * pyramid + expansion.  The first call after rcflow_stream_reset only primes the state
 * and writes no flow (returns 1 instead of RC_OK); so does the first call with a different
 * frame size or different parameters (the cached expansion belongs to the old ones). */
int rcflow_push_frame_dev(rc_ctx* ctx, int stream, const uint8_t* d_frame, size_t step,
                          int w, int h, float* d_flow_xy, size_t flow_step,
                          const rc_farneback_params* p);
/* The same loop with HOST frames, as the reference holds them after video.read / resize / cvtColor
 * (ripcurrents.cpp:198-213): the frame is copied into one of two page-locked staging buffers and uploaded
 * asynchronously; the call returns once the copy into the staging buffer is done, so upload and kernels of frame t
 * overlap the host's decode of frame t + 1.  The flow field stays on the device (rcflow_stream_flow_ptr: input of the
 * analysis entry points) and crosses PCIe only through rcflow_stream_flow_read.  Returns 1 when the call primed the
 * stream.  Interoperates with rcflow_push_frame_dev / rcflow_push_clip_dev on the same slot. */
int rcflow_push_frame_u8(rc_ctx* ctx, int stream, const uint8_t* frame, size_t step, int w, int h,
                         const rc_farneback_params* p);
/* Without the copy: rcflow_frame_buffer_acquire hands out the next of the slot's two page-locked staging buffers (w x h
 * bytes, dense rows; it waits until the upload that last read that buffer has left it) for the host to produce the frame
 * INTO -- e.g. as the destination Mat of the cvtColor at ripcurrents.cpp:210 -- and rcflow_push_frame_acquired pushes it
 * like rcflow_push_frame_u8 does (same return values, asynchronous).  One buffer is out at a time: acquiring again, or a
 * rcflow_push_frame_u8 on the slot, takes the same buffer back (RC_ESTATE from a push without an acquisition). */
int rcflow_frame_buffer_acquire(rc_ctx* ctx, int stream, int w, int h, uint8_t** host_frame, size_t* step);
int rcflow_push_frame_acquired(rc_ctx* ctx, int stream, const rc_farneback_params* p);
int rcflow_stream_flow_ptr(rc_ctx* ctx, int stream, float** d_flow_xy, int* w, int* h);
int rcflow_stream_flow_read(rc_ctx* ctx, int stream, float* flow_xy, size_t flow_step);

/* One whole iteration of the reference's frame loop (ripcurrents.cpp:194-479) per call, on the device (with use_graph:
 * as ONE hipGraph launch per frame once the slot has seen the same configuration twice): the frame the host produced into the buffer of
 * rcflow_frame_buffer_acquire is uploaded and expanded, the flow against the previous frame is computed (:215), then on
 * that resident field: streamline_field(dt, iterations) with the PREVIOUS frame's UPPER (:229-231), the seed
 * streamlines (:283-285, rcflow_advect_points_dev semantics; nseeds may be 0), the cumulative histogram and the
 * thresholds (:319-366), create_flow + create_accumulationbuffer with framecount = the number of flow fields of this
 * stream so far (:376-439; the counter lives on the device so that the captured launch sequence stays valid), and the
 * mask's edges (:477-479; d_edges and d_outmask may be NULL).  Same kernels, same order, same bits as the separate
 * calls.  Returns 1 when the call only primed the stream (first frame, or another size / other parameters: no flow
 * yet), RC_ESTATE without an acquired frame buffer.  The slot keeps a ring of two expansions while it is driven this
 * way; rcflow_stream_reset (or any other entry point on the slot) restarts it. */
typedef struct rc_frame_loop {
    float dt; int iterations;                 /* streamline_field */
    float* d_seeds; int nseeds;               /* device, nseeds x (x, y), advanced in place */
    int seed_variant; float seed_dt; int seed_iterations; float seed_upper;
    float MID, LOWER;                         /* ripcurrents.cpp:142-143: 0.5, 0.2 */
    uint8_t* d_outmask; size_t mask_step;     /* device 8UC1, optional */
    uint8_t* d_edges; size_t edges_step;      /* device 8UC1, optional (needs d_outmask) */
    int use_graph;                            /* 0 (default): the launches are issued one by one, upload and expansion on the slot's
                                               * second stream; 1: one captured hipGraph launch per frame on a ring of two expansions */
} rc_frame_loop;
int rcflow_frame_loop_step(rc_ctx* ctx, int stream, const rc_farneback_params* p, const rc_frame_loop* loop);
/* Batched form of the same stream: the nframes frames continue the slot's stream, every frame is expanded
 * once however the segment is cut into calls.  Returns the number of flow fields written to d_flows[0..):
 * nframes if the stream was primed (flow 0 = last frame of the previous call -> d_frames[0]), nframes - 1 if
 * this call primed it; negative RC_E* on error.  Interoperates with rcflow_push_frame_dev. */
int rcflow_push_clip_dev(rc_ctx* ctx, int stream, const uint8_t* d_frames, size_t frame_stride, size_t step,
                         int nframes, int w, int h, float* d_flows_xy, size_t flow_frame_stride,
                         size_t flow_step, const rc_farneback_params* p);
int rcflow_stream_reset(rc_ctx* ctx, int stream);
/* A whole resident clip: nframes frames -> nframes-1 flow fields (pair t = frames t,t+1),
 * processed `chunk` pairs per launch.  The bench path. */
int rcflow_farneback_clip_dev(rc_ctx* ctx, int stream, const uint8_t* d_frames,
                              size_t frame_stride, size_t step, int nframes, int w, int h,
                              float* d_flows_xy, size_t flow_frame_stride, size_t flow_step,
                              const rc_farneback_params* p);
/* Lockstep batch of `nstreams` independent video streams (BASELINE config 5): frame t of every
 * stream arrives together as d_frames[nstreams][h][w]; the slot keeps every stream's previous
 * expansion, so each call is one pyramid + expansion + flow per stream, all streams in the same
 * launches.  The first call after rcflow_batch_reset primes the state and returns 1.  With
 * use_graph != 0 the launch sequence is captured into a hipGraph the second time the same
 * buffers are seen and replayed from then on (two graphs, one per ring parity). */
int rcflow_push_batch_dev(rc_ctx* ctx, int stream, const uint8_t* d_frames, size_t frame_stride,
                          size_t step, int nstreams, int w, int h, float* d_flows_xy,
                          size_t flow_frame_stride, size_t flow_step,
                          const rc_farneback_params* p, int use_graph);
int rcflow_batch_reset(rc_ctx* ctx, int stream);
/* Level geometry actually used (levels cropped at min_size 32, cvRound sizes).
 * Returns the cropped `levels`; scales are k = 0..levels. */
int rcflow_level_geometry(int w, int h, double pyr_scale, int levels, int k, int* wk, int* hk);

/* Stage-level entry points (device pointers; used by the parity tests to compare each
 * kernel with the oracle stage by stage; interleaved layouts as in OpenCV). */
int rcflow_stage_pyr_level_dev(rc_ctx* ctx, int stream, const uint8_t* d_img, size_t step,
                               int w, int h, double pyr_scale, int k, float* d_out /* hk*wk */);
int rcflow_stage_polyexp_dev(rc_ctx* ctx, int stream, const float* d_I, int w, int h,
                             int poly_n, double poly_sigma, float* d_R5 /* h*w*5 */);
/* One FarnebackUpdateMatrices + FarnebackUpdateFlow_* application:
 * flow_out = solve(blur(M(R0, R1, flow_in))) */
int rcflow_stage_flow_iter_dev(rc_ctx* ctx, int stream, const float* d_R0_5,
                               const float* d_R1_5, const float* d_flow_in, int w, int h,
                               int winsize, int flags, float* d_flow_out);

/* ------------------------------------------------------------------ B: analysis
 * Per-slot device-resident state mirrors the locals of ripcurrents.cpp:133-176:
 * hist/hist2d/histsum/histsum2d (cumulative, never reset by the reference), UPPER (=100
 * initially), UPPER2d, prop_above_upper, accumulator, streamlines_mat/_distance. */
int rcflow_analysis_reset(rc_ctx* ctx, int stream, int w, int h);

/* Replaces the counting loop of create_histogram (ripcurrents_module.cpp:94-107,
 * ripcurrents.cpp:319-330) fused with the polar conversion ripcurrents.cpp:305-309.
 * Adds this flow field's counts to the slot's cumulative histogram. */
int rcflow_histogram_dev(rc_ctx* ctx, int stream, const float* d_flow_xy, size_t flow_step,
                         int w, int h);
/* The counters are int32 like the reference's (`int hist[50]`, ripcurrents.cpp:147-150, which wrap after
 * 2^31 / (w*h) frames): a call that could carry histsum past INT32_MAX returns RC_ESTATE and counts nothing;
 * rcflow_histogram_reset_dev starts a new segment (asynchronous zero of the counters only). */
int rcflow_histogram_reset_dev(rc_ctx* ctx, int stream);
/* The same for `count` resident flow fields in one launch (a segment's flows). */
int rcflow_histogram_clip_dev(rc_ctx* ctx, int stream, const float* d_flows_xy,
                              size_t flow_frame_stride, size_t flow_step, int count, int w, int h);
/* Replaces the threshold scans of create_histogram (ripcurrents_module.cpp:109-144):
 * derives UPPER, UPPER2d[36], prop_above_upper[36] on the device from the slot's counts. */
int rcflow_thresholds_dev(rc_ctx* ctx, int stream);
/* the same scans on a caller-held block of RC_HIST_WORDS device counters (the all-reduced global
 * histogram); the slot's own cumulative counters are not touched */
int rcflow_thresholds_words_dev(rc_ctx* ctx, int stream, const int32_t* d_words);
/* Copies the slot's histogram words (RC_HIST_WORDS int32: hist, hist2d, histsum,
 * histsum2d) and thresholds to the host; any pointer may be NULL.  Synchronises. */
int rcflow_histogram_read(rc_ctx* ctx, int stream, int32_t* hist, int32_t* hist2d,
                          int32_t* histsum, int32_t* histsum2d, float* UPPER, float* UPPER2d,
                          float* prop_above_upper);
int rcflow_histogram_write(rc_ctx* ctx, int stream, const int32_t* words /* RC_HIST_WORDS */);
/* Device address of the RC_HIST_WORDS int32 block (for an RCCL all-reduce issued by the
 * host framework on the slot's stream). */
int rcflow_histogram_device_ptr(rc_ctx* ctx, int stream, int32_t** d_words);

/* Replaces create_flow + create_accumulationbuffer (ripcurrents_module.cpp:153-212,
 * ripcurrents.cpp:376-439) fused with the polar conversion.  Uses the slot's UPPER /
 * UPPER2d; MID/LOWER are ripcurrents.cpp:142-143.  Optional outputs (device, may be
 * NULL): polar_hsv 32FC3 (angle, sat, val rescaled: the `current` the reference displays),
 * waterclass 32FC3, out 32FC3, outmask 8UC1.  The slot's accumulator (.x channel of the
 * reference's 32FC3 accumulator) is updated when framecount > 30. */
int rcflow_classify_accumulate_dev(rc_ctx* ctx, int stream, const float* d_flow_xy,
                                   size_t flow_step, int w, int h, int framecount, float MID,
                                   float LOWER, float* d_polar_hsv, size_t polar_step,
                                   float* d_waterclass, size_t wc_step, float* d_out,
                                   size_t out_step, uint8_t* d_outmask, size_t mask_step);
int rcflow_accumulator_read(rc_ctx* ctx, int stream, float* acc /* h*w */);

/* Replaces streamlines_mat.forEach(streamline_field(...)) ripcurrents.cpp:229-231
 * (ripcurrents_module.cpp:608-648): one particle per pixel, state in the slot.
 * UPPER < 0 means "use the slot's current UPPER" (the value the previous frame's
 * histogram produced, as in the reference's call order).  iterations: 0..65536 (the reference
 * passes 1 or 100), RC_EINVAL beyond -- the loop runs on the device. */
int rcflow_advect_field_dev(rc_ctx* ctx, int stream, const float* d_flow_xy, size_t flow_step,
                            int w, int h, float dt, int iterations, float UPPER);
int rcflow_advect_field_read(rc_ctx* ctx, int stream, float* pt_xy /* h*w*2 */,
                             float* dist /* h*w */);
/* Replaces the seed loops over streamline()/streamline_2()/streamline_3()/pathlines
 * (ripcurrents.cpp:283-285, ripcurrents_module.cpp:72-75): variants
 * 0 ripcurrents_module.cpp:486-528, 1 :531-569, 2 :572-606, 3 ripcurrents.cpp:656-698,
 * 4 pathlines.cpp:9-46.  d_pts is n x (x,y), updated in place; d_trace (optional)
 * receives the position after every step (n*iters*2 floats) for the host to draw. */
int rcflow_advect_points_dev(rc_ctx* ctx, int stream, float* d_pts, int n,
                             const float* d_flow_xy, size_t flow_step, int w, int h, float dt,
                             int iterations, float UPPER, int variant, float* d_trace);
/* get_delta over every pixel (ripcurrents_module.cpp:395-397,650-679). */
int rcflow_get_delta_field_dev(rc_ctx* ctx, int stream, float* d_pt_xy, size_t pt_step,
                               const float* d_flow_xy, size_t flow_step, int w, int h,
                               float dt, float UPPER);

/* Flow-field post-ops (in place on device flow fields) */
int rcflow_subtract_average_dev(rc_ctx* ctx, int stream, float* d_flow_xy, size_t flow_step,
                                int w, int h);                 /* ripcurrents_module.cpp:810-898 */
int rcflow_subtract_mean_magnitude_dev(rc_ctx* ctx, int stream, float* d_flow_xy,
                                       size_t flow_step, int w, int h);   /* :900-1015 */
int rcflow_stabilizer_dev(rc_ctx* ctx, int stream, float* d_flow_xy, size_t flow_step, int w,
                          int h);                              /* :279-308 */
int rcflow_window_mean_dev(rc_ctx* ctx, int stream, float* d_avg, float* d_slot,
                           const float* d_cur, size_t n, int window);     /* main.cpp:1142-1153 */
/* Colouring: HSV triples as the reference builds them before cvtColor(HSV2BGR). */
int rcflow_vector_to_color_dev(rc_ctx* ctx, int stream, const float* d_flow_xy,
                               size_t flow_step, int w, int h, uint8_t* d_hsv, size_t hsv_step,
                               float* max_displacement_io);    /* :1017-1057 */
int rcflow_shear_rate_to_color_dev(rc_ctx* ctx, int stream, const float* d_flow_xy,
                                   size_t flow_step, int w, int h, uint8_t* d_hsv,
                                   size_t hsv_step, float* max_frobenius_io); /* :1059-1138 */

/* ------------------------------------------------------------------ SURVEY 8(f) "next" rows
 * create_edges(outmask) ripcurrents_module.cpp:216-220 (ripcurrents.cpp:477-479): 5x5
 * MORPH_ELLIPSE dilate followed by the morphological gradient, fused.  Not in place. */
int rcflow_create_edges_dev(rc_ctx* ctx, int stream, const uint8_t* d_outmask, size_t mask_step,
                            int w, int h, uint8_t* d_edges, size_t edges_step);
/* create_output(subframe, outmask) ripcurrents_module.cpp:225-244 (ripcurrents.cpp:487-505): in place,
 * red channel of the 8UC3 frame := 255 wherever the edge mask is non-zero. */
int rcflow_create_output_dev(rc_ctx* ctx, int stream, uint8_t* d_subframe_bgr, size_t step,
                             const uint8_t* d_outmask, size_t mask_step, int w, int h);
/* resize(frame, subframe, Size(dw,dh), 0, 0, INTER_LINEAR) + cvtColor(COLOR_BGR2GRAY)
 * (ripcurrents.cpp:209-210, main.cpp:258-259): 8UC3 BGR frame in, 8UC1 out. */
int rcflow_resize_bgr_to_gray_dev(rc_ctx* ctx, int stream, const uint8_t* d_bgr, size_t step,
                                  int sw, int sh, uint8_t* d_gray, size_t gray_step, int dw, int dh);
/* The same with INTER_AREA, as the reference resizes the FIRST frame of a run (ripcurrents.cpp:186,
 * main.cpp:126, 223, ...); shrinking only. */
int rcflow_resize_area_bgr_to_gray_dev(rc_ctx* ctx, int stream, const uint8_t* d_bgr, size_t step,
                                       int sw, int sh, uint8_t* d_gray, size_t gray_step, int dw, int dh);

/* Display path, ripcurrents.cpp:233-273 (= streamline_displacement / _total_motion / _ratio /
 * _positions, ripcurrents_module.cpp:13-60) on the slot's streamline field (rcflow_advect_field_dev):
 * which 0 = |pt|, 1 = dist, 2 = |pt| / dist; minMaxLoc + convertTo(CV_8UC1, 255/max) +
 * applyColorMap(COLORMAP_JET) -> 8UC3 BGR.  max_out (host, may be NULL) receives the maximum (blocks). */
int rcflow_streamline_display_dev(rc_ctx* ctx, int stream, int which, uint8_t* d_bgr, size_t bgr_step,
                                  float* max_out);
/* marks (1,1,1) in a 32FC3 image where each pixel's particle sits (:44-60); the caller zeroes it */
int rcflow_streamline_positions_dev(rc_ctx* ctx, int stream, float* d_density, size_t density_step);
/* cvtColor(current, current, CV_HSV2BGR) on the 32FC3 display image (ripcurrents.cpp:405); H in degrees */
int rcflow_hsv_to_bgr_dev(rc_ctx* ctx, int stream, const float* d_hsv, size_t hsv_step, int w, int h,
                          float* d_bgr, size_t bgr_step);
/* frame size of the slot's analysis state (0, 0 before rcflow_analysis_reset / the first analysis call) */
int rcflow_analysis_size(rc_ctx* ctx, int stream, int* w, int* h);
/* the 256 x BGR table of applyColorMap(COLORMAP_JET) (host) */
int rcflow_jet_lut(uint8_t* lut_bgr /* 768 */);

/* Sparse pyramidal Lucas-Kanade: cv::calcOpticalFlowPyrLK(prev, next, prevPts, nextPts, status, err,
 * winSize, maxLevel, criteria, flags, minEigThreshold) on 8UC1 device images -- Streakline.cpp:32,
 * ripcurrents_module.cpp:716, :738, :775, :1162.  d_prev_pts / d_next_pts: npts x (x, y) floats on the
 * device; d_status npts bytes; d_err npts floats or NULL.  crit_type bit 0 = TermCriteria::COUNT,
 * bit 1 = EPS; flags: 4 = OPTFLOW_USE_INITIAL_FLOW (d_next_pts is then also an input), 8 =
 * OPTFLOW_LK_GET_MIN_EIGENVALS.  maxLevel < 8, windows up to 128x128. */
int rcflow_pyrlk_dev(rc_ctx* ctx, int stream, const uint8_t* d_prev, size_t prev_step,
                     const uint8_t* d_next, size_t next_step, int w, int h, const float* d_prev_pts,
                     float* d_next_pts, int npts, uint8_t* d_status, float* d_err, int win_w, int win_h,
                     int max_level, int crit_type, int max_count, double epsilon, int flags,
                     double min_eig_threshold);
/* the same with host pointers (the cv:: signature's form): copies in, tracks, copies out, blocking */
int rcflow_pyrlk_u8(rc_ctx* ctx, int stream, const uint8_t* prev, size_t prev_step, const uint8_t* next,
                    size_t next_step, int w, int h, const float* prev_pts, float* next_pts, int npts,
                    uint8_t* status, float* err, int win_w, int win_h, int max_level, int crit_type,
                    int max_count, double epsilon, int flags, double min_eig_threshold);
/* last pyramid level buildOpticalFlowPyramid keeps for this size and window */
int rcflow_pyrlk_levels(int w, int h, int win_w, int win_h, int max_level);

/* ------------------------------------------------------------------ multi-GPU: the global flow histogram
 * one process per GPU, each on its own video segment; the only exchange is the integer sum of
 * the RC_HIST_WORDS histogram counters (7548 B) over RCCL, after which every rank derives the same global
 * UPPER / UPPER2d / prop_above_upper (what ripcurrents.cpp:333-366 computes from one stream's counters).
 * librccl is opened at run time by these calls only.  A world of one rank with a NULL id is the identity and needs no
 * RCCL; with an id it is a one-rank RCCL communicator like any other. */
#define RC_COMM_ID_BYTES 128   /* sizeof(ncclUniqueId) */
/* rank 0: creates the id; the host distributes it to the other ranks (MPI, a socket, a file) */
int rcflow_comm_unique_id(void* id_out /* RC_COMM_ID_BYTES */);
/* every rank, collectively: joins the communicator on the context's GPU.  RC_ECOMM when RCCL fails. */
int rcflow_comm_init(rc_ctx* ctx, const void* unique_id /* RC_COMM_ID_BYTES; NULL when world == 1: no RCCL */,
                     int rank, int world);
int rcflow_comm_destroy(rc_ctx* ctx);
int rcflow_comm_rank(rc_ctx* ctx, int* rank, int* world);
/* Starts the sum over all ranks of the slot's histogram counters as they are at this point of the slot's
 * stream; the result goes to d_words_out (device, RC_HIST_WORDS int32) or, when NULL, to a context-owned
 * buffer (rcflow_allreduce_hist_result).  Asynchronous and off the slot's stream: the collective runs on its
 * own HIP stream beside whatever the slot does next.  Collective: every rank calls it the same number of times,
 * and no rank ever skips the reduction on a local condition (a rank that did would leave the others waiting). */
int rcflow_allreduce_hist(rc_ctx* ctx, int stream, int32_t* d_words_out);
/* Verdict on the collective started last, the same on every rank (it comes from a reduced word): waits for the
 * collective on the host, then RC_ESTATE when the ranks together counted more pixels than the int32 histsum of the
 * reference (ripcurrents.cpp:147-150) can hold -- reduce per shorter segment -- else RC_OK.  pixels_counted (may be
 * NULL): the upper bound used. */
int rcflow_allreduce_hist_status(rc_ctx* ctx, long long* pixels_counted);
/* Orders the slot's stream after the collective started last (no host wait): call it before
 * rcflow_thresholds_words_dev(ctx, stream, d_words_out). */
int rcflow_allreduce_hist_join(rc_ctx* ctx, int stream);
int rcflow_allreduce_hist_result(rc_ctx* ctx, int32_t** d_words);

/* ------------------------------------------------------------------ measurement */
/* When enabled every kernel launch is bracketed by HIP events on the slot's stream.  Measurement
 * aid: while it is on, drive the context from one thread only (the event list is per context). */
int rcflow_profile_enable(rc_ctx* ctx, int on);
int rcflow_profile_reset(rc_ctx* ctx);
/* Resolves pending events and returns per-kernel totals.  names[i] points to a static
 * string "kernel@level"; returns the number of entries written (<= cap).  alg_bytes = the
 * launches' compulsory bytes (inputs once + outputs once); model_bytes = SURVEY.md section
 * 8(d)'s algorithmic bytes of the stages those launches stand for. */
int rcflow_profile_read(rc_ctx* ctx, int cap, const char** names, int* launches,
                        double* total_ms, double* alg_bytes, double* model_bytes);

/* The same totals under the reference's own bucket names, in the order it prints them (ripcurrents.cpp:103-109,
 * :518-524): farneback, polar, threshold, overlay, erosion, codec, stream ("pathlines").  GPU time of the kernels
 * that do each bucket's work; "polar" is 0 (the cartToPolar of :305-309 is fused into the histogram and
 * classification kernels, booked under "threshold"), "codec" is 0 (video decode is host I/O outside the library).
 * names / ms: RC_PROFILE_BUCKETS entries each (either may be NULL).  Returns RC_PROFILE_BUCKETS. */
#define RC_PROFILE_BUCKETS 7
int rcflow_profile_read_buckets(rc_ctx* ctx, const char** names, double* ms);

#ifdef __cplusplus
}
#endif
#endif
