/*
 * rc_oracle.h -- C interface of the CPU oracle (TEST INFRASTRUCTURE, NOT PRODUCT).
 *
 * PARITY UNPINNED: the reference (borgor/ripcurrents) holds no golden vectors,
 * fixtures or tests for this path, and the Farneback arithmetic lives in
 * un-vendored OpenCV 4.1.0 (modules/video/src/optflow.cpp, imgproc smooth/resize,
 * core mathfuncs), absent from /root/reference and from this image.  This oracle
 * restates those published algorithms; it is pinned only by the analytic
 * known-answer tests in tests/ (see DESIGN.md "Oracle").
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
 * this library.  The product (librcflow.so) never links or calls it.
 *
 * Layouts follow cv::Mat: interleaved channels, row pointer + byte step.
 */
#ifndef RC_ORACLE_H
#define RC_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_FARNEBACK_GAUSSIAN 256 /* cv::OPTFLOW_FARNEBACK_GAUSSIAN */
#define ORC_HIST_BINS 50           /* ripcurrents.hpp:7 */
#define ORC_HIST_DIRECTIONS 36     /* ripcurrents.hpp:8 */
#define ORC_HIST_RESOLUTION 20     /* ripcurrents.hpp:9 */

/* ---- A rows: Farneback (OpenCV 4.1.0 CPU path; call site ripcurrents.cpp:215) ---- */

/* cv::calcOpticalFlowFarneback(prev,next,flow,...) on 8UC1 inputs, CV_32FC2 out.
 * nthreads>1 row-stripes the stages that have no cross-row state. */
int orc_farneback_u8(const uint8_t* prev, size_t prev_step, const uint8_t* next,
                     size_t next_step, int w, int h, float* flow, size_t flow_step,
                     double pyr_scale, int levels, int winsize, int iters, int poly_n,
                     double poly_sigma, int flags, int nthreads);

/* The same call with diagnostics for the parity tests (the tolerance is stated on
 * pixels whose 2x2 system is well conditioned).  Any pointer may be NULL.
 *   g_last      h*w*3  the window values (g11, g12, g22) of the LAST iteration at scale 0, i.e.
 *                      the matrix the final solve inverted (box window: already scaled by 1/winsize^2)
 *   det_min     h*w    min over every scale and iteration of g11*g22 - g12^2 over the 3x3 block
 *                      around the pixel's ancestor at that scale (conditioning of the whole
 *                      coarse-to-fine path that produced the pixel)
 *   level_flow  [k]    w_k*h_k*2: the flow at the end of scale k (before up-sampling) */
#define ORC_MAX_DIAG_LEVELS 12
typedef struct orc_farneback_diag {
    float* g_last;
    float* det_min;
    float* level_flow[ORC_MAX_DIAG_LEVELS];
} orc_farneback_diag;
int orc_farneback_u8_ex(const uint8_t* prev, size_t prev_step, const uint8_t* next,
                        size_t next_step, int w, int h, float* flow, size_t flow_step,
                        double pyr_scale, int levels, int winsize, int iters, int poly_n,
                        double poly_sigma, int flags, int nthreads, const orc_farneback_diag* diag);

/* number of pyramid scales actually used (levels cropped by min_size=32) and
 * the size of scale k; returns cropped `levels` (scales are k=0..levels). */
int orc_level_geometry(int w, int h, double pyr_scale, int levels, int k, int* wk,
                       int* hk, double* sigma, int* ksize);

/* A1: convertTo(32F) -> GaussianBlur(ksize,sigma) -> resize(INTER_LINEAR) */
int orc_pyr_level(const uint8_t* img, size_t step, int w, int h, double sigma, int ksize,
                  float* out, int ow, int oh);
/* building blocks of A1 */
int orc_gaussian_kernel(int n, double sigma, float* k);
int orc_gaussian_blur_f32(const float* src, int w, int h, int ksize, double sigma, float* dst);
int orc_resize_linear_f32(const float* src, int sw, int sh, int cn, float* dst, int dw, int dh);

/* A2: FarnebackPrepareGaussian / FarnebackPolyExp. g/xg/xxg have 2n+1 entries
 * (index n is tap 0); ig = {ig11, ig03, ig33, ig55}. R is h*w*5 interleaved. */
int orc_prepare_gaussian(int n, double sigma, float* g, float* xg, float* xxg, double* ig);
int orc_polyexp(const float* I, int w, int h, int n, double sigma, float* R);

/* A3: FarnebackUpdateMatrices over rows [y0,y1) */
int orc_update_matrices(const float* R0, const float* R1, const float* flow, float* M,
                        int w, int h, int y0, int y1);
/* A4/A5 (+A6 stripe logic): FarnebackUpdateFlow_Blur / _GaussianBlur */
int orc_update_flow_blur(const float* R0, const float* R1, float* flow, float* M, int w,
                         int h, int block_size, int update_matrices);
int orc_update_flow_gaussian(const float* R0, const float* R1, float* flow, float* M,
                             int w, int h, int block_size, int update_matrices);

/* ---- B rows: per-pixel analysis (arithmetic in the reference itself) ---- */

/* B1 ripcurrents.cpp:305-309: cartToPolar(x,y,mag,angle,degrees) -> (angle,mag,mag) */
void orc_fast_atan2_deg(const float* y, const float* x, float* angle, int n);
void orc_flow_to_polar(const float* flow, size_t flow_step, int w, int h, float* polar,
                       size_t polar_step);

/* B2 ripcurrents_module.cpp:89-107 (counts) and :109-144 (thresholds).
 * hist2d is [36][50] row-major. A direction index of 36 (angle==360.0f, an
 * out-of-bounds write in the reference) is folded into direction 0. */
void orc_histogram_accumulate(const float* polar, size_t polar_step, int w, int h,
                              int32_t* hist, int32_t* histsum, int32_t* hist2d,
                              int32_t* histsum2d);
void orc_histogram_thresholds(const int32_t* hist, int32_t histsum, const int32_t* hist2d,
                              const int32_t* histsum2d, float* UPPER, float* UPPER2d,
                              float* prop_above_upper);

/* B3a ripcurrents_module.cpp:153-182 create_flow */
void orc_create_flow(float* polar, size_t polar_step, float* waterclass, size_t wc_step,
                     float* accumulator2, size_t acc2_step, int w, int h, float UPPER,
                     float MID, float LOWER, const float* UPPER2d);
/* B3b ripcurrents_module.cpp:189-212 create_accumulationbuffer */
void orc_create_accumulationbuffer(float* accumulator, size_t acc_step,
                                   const float* accumulator2, size_t acc2_step, float* out,
                                   size_t out_step, uint8_t* outmask, size_t mask_step,
                                   int w, int h, int framecount);

/* B4 ripcurrents_module.cpp:608-648 streamline_field over every pixel */
void orc_streamline_field(float* pt, size_t pt_step, float* dist, size_t dist_step,
                          const float* flow, size_t flow_step, int w, int h, float dt,
                          int iterations, float UPPER);
/* B5: seed advection. variant 0 = ripcurrents_module.cpp:486-528 (step delta*dt,
 * cutoff r>UPPER); 1 = :531-569 (delta*dt, cutoff r>5); 2 = :572-606 (100 fixed
 * iterations, delta*0.1, no cutoff); 3 = ripcurrents.cpp:656-698 (delta*dt/iters,
 * cutoff r>UPPER); 4 = pathlines.cpp:9-46 (delta*dt/iters, no cutoff).
 * pts is n x (x,y). trace (optional) receives n*iters*2 positions after each step
 * (unchanged position repeated once a particle has stopped). */
void orc_streamline_points(float* pts, int n, const float* flow, size_t flow_step, int w,
                           int h, float dt, int iterations, float UPPER, int variant,
                           float* trace);
/* get_delta ripcurrents_module.cpp:650-679 over every pixel (averageVector :395-397) */
void orc_get_delta_field(float* pt, size_t pt_step, const float* flow, size_t flow_step,
                         int w, int h, float dt, float UPPER);

/* B6 Streakline.cpp:22-71 bookkeeping with the vertices moved through a dense
 * flow field (main.cpp:961-977 precedent) instead of PyrLK.  verts holds
 * *nverts (x,y) pairs, capacity >= *nverts+1. */
void orc_streakline_step(float* verts, int* nverts, float gen_x, float gen_y,
                         const float* flow, size_t flow_step, int w, int h, float dt,
                         int* frame_count);

/* B7 post-ops */
void orc_subtract_average(float* flow, size_t flow_step, int w, int h);          /* :810-898 */
void orc_subtract_mean_magnitude(float* flow, size_t flow_step, int w, int h);   /* :900-1015 */
void orc_stabilizer(float* flow, size_t flow_step, int w, int h);                /* :279-308 */
void orc_window_mean_update(float* avg, float* slot, const float* cur, int n,    /* main.cpp:1142-1153 */
                            int window);

/* B8 colouring (HSV triples, before the cvtColor display step) */
void orc_vector_to_color(const float* flow, size_t flow_step, int w, int h, uint8_t* hsv,
                         size_t hsv_step, float* max_displacement);               /* :1017-1057 */
void orc_shear_rate_to_color(const float* flow, size_t flow_step, int w, int h,
                             uint8_t* hsv, size_t hsv_step, float* max_frobenius); /* :1059-1138 */

/* ---- section 8(f) "next" rows ---- */
/* create_edges ripcurrents_module.cpp:216-220 (ripcurrents.cpp:477-479): 5x5 MORPH_ELLIPSE
 * dilate, then morphological gradient (dilate - erode) of the result.  OpenCV imgproc
 * morph.cpp semantics: constant border that never wins (0 for dilate, 255 for erode). */
void orc_create_edges(const uint8_t* mask, size_t mask_step, int w, int h, uint8_t* out,
                      size_t out_step);
void orc_ellipse5(uint8_t kernel[25]);   /* getStructuringElement(MORPH_ELLIPSE, Size(5,5)) */
/* Frame pre-processing ripcurrents.cpp:209-210: resize(frame, subframe, Size(dw,dh), 0, 0,
 * INTER_LINEAR) on 8UC3 (imgproc resize.cpp fixed-point path, 11-bit coefficients) followed by
 * cvtColor(COLOR_BGR2GRAY) (color_rgb.cpp: (B*1868 + G*9617 + R*4899 + 8192) >> 14). */
void orc_resize_bgr_to_gray(const uint8_t* bgr, size_t step, int sw, int sh, uint8_t* gray,
                            size_t gray_step, int dw, int dh);

/* The first frame's pre-processing (ripcurrents.cpp:186, main.cpp:126, ...): resize INTER_AREA on 8UC3
 * (imgproc resize.cpp resizeAreaFast_ for integer factors, resizeArea_ otherwise) + cvtColor(BGR2GRAY). */
void orc_resize_area_bgr_to_gray(const uint8_t* bgr, size_t step, int sw, int sh, uint8_t* gray,
                                 size_t gray_step, int dw, int dh);

/* Display path (section 8(f) row 4; display_oracle.cpp): ripcurrents.cpp:233-273 and :405 */
void orc_jet_lut(uint8_t* lut_bgr /* 256*3: applyColorMap(COLORMAP_JET) */);
/* which 0 = streamline_displacement, 1 = streamline_total_motion, 2 = streamline_ratio
 * (ripcurrents_module.cpp:13-40): minMaxLoc + convertTo(8U, 255/max) + JET -> 8UC3 BGR */
void orc_streamline_display(const float* pt, size_t pt_step, const float* dist, size_t dist_step, int w,
                            int h, int which, uint8_t* bgr, size_t bgr_step, double* max_out);
void orc_streamline_positions(const float* pt, size_t pt_step, int w, int h, float* density,
                              size_t density_step);                       /* :44-60, 32FC3 */
void orc_hsv_to_bgr_f32(const float* hsv, size_t hsv_step, int w, int h, float* bgr, size_t bgr_step);

/* Sparse pyramidal Lucas-Kanade (section 8(f) row 3; lk_oracle.cpp): cv::calcOpticalFlowPyrLK on
 * 8UC1 images, scalar path of OpenCV 4.1.0 lkpyramid.cpp.  pts are n x (x, y); crit_type bit 0 =
 * COUNT, bit 1 = EPS (cv::TermCriteria); flags: 4 = OPTFLOW_USE_INITIAL_FLOW, 8 =
 * OPTFLOW_LK_GET_MIN_EIGENVALS.  Call sites Streakline.cpp:32, ripcurrents_module.cpp:716,738,775,1162. */
int orc_pyrlk(const uint8_t* prev, size_t prev_step, const uint8_t* next, size_t next_step, int w,
              int h, const float* prev_pts, float* next_pts, int npts, uint8_t* status, float* err,
              int win_w, int win_h, int max_level, int crit_type, int max_count, double epsilon,
              int flags, double min_eig_threshold);
int orc_pyrlk_levels(int w, int h, int win_w, int win_h, int max_level);  /* last level actually used */
int orc_pyrdown_u8(const uint8_t* src, size_t step, int w, int h, uint8_t* dst, size_t dst_step);
int orc_scharr_deriv(const uint8_t* src, size_t step, int w, int h, int16_t* dxy);  /* h*w*2 (dx, dy) */
/* Streakline::runLK (Streakline.cpp:22-71) with the reference's own mover (PyrLK 50x50, 3 levels) */
int orc_streakline_step_lk(float* verts, int* nverts, float gen_x, float gen_y, const uint8_t* prev,
                           size_t prev_step, const uint8_t* next, size_t next_step, int w, int h,
                           int* frame_count);

#ifdef __cplusplus
}
#endif
#endif
