// rc_host.h -- host-side context, plans and stream slots of librcflow (internal).
#pragma once

#include <string>
#include <vector>

#include "rc_common.h"

struct RcBuf {
    void* p = nullptr;
    size_t bytes = 0;
};

struct RcPlan {
    bool valid = false;
    int w = 0, h = 0;
    rc_farneback_params prm{};
    int nlev = 0;  // number of scales = cropped levels + 1
    RcLevel lv[RC_MAX_LEVELS];
    size_t kern_off[RC_MAX_LEVELS];
    RcPolyK pk;
    RcWindow win;
    int nslots = 0, chunk = 0;
    int exact_taps = 0;
    int exact = 0;
};

// Device-resident analysis state of one stream slot (ripcurrents.cpp:133-176).
struct RcAnalysis {
    int w = 0, h = 0;
    RcBuf hist;        // RC_HIST_WORDS int32
    long long hist_added = 0;   // pixels passed to the histogram since the last reset (upper bound of histsum)
    RcBuf hist_part;   // partial hist2d tables of the histogram kernel
    RcBuf thr;         // UPPER | UPPER2d[36] | prop[36] floats (+1 pad)
    RcBuf acc;         // h*w float accumulator (.x channel of the reference's 32FC3)
    RcBuf pt;          // h*w float2 streamlines_mat
    RcBuf dist;        // h*w float streamlines_distance
    RcBuf scratch;     // reductions
    RcBuf jet;         // COLORMAP_JET LUT (768 B) + the display maximum key
    RcBuf loopc;       // one int32: framecount of rcflow_frame_loop_step (incremented on the device)
};

struct RcBatchKey {
    const void* frames; size_t frame_stride, step;
    void* flows; size_t flow_frame_stride, flow_step;
    void* hip_stream;
};

struct RcSlot {
    hipStream_t own = nullptr, cur = nullptr;
    hipStream_t aux = nullptr;  // second stream: expansions beside flow kernels (clip path option "overlap"; frame loop)
    // frame loop on two streams (rcflow_push_frame_dev): events in a small ring, reused in FIFO order
    hipEvent_t fev[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    int fev_i = 0;
    hipEvent_t flow_done[2] = {nullptr, nullptr};   // the flow launches of the last two pushes
    int flow_done_i = 0;
    // consecutive two-stream pushes so far: the second stream may run ahead of the slot's stream only past work those
    // pushes recorded themselves; anything else queued on the slot (a clip, a pair call, priming, the one-stream path)
    // zeroes the streak (expand_frames / compute_flows) and the next two-stream push first joins the slot's stream
    int ts_streak = 0;
    bool in_ts_push = false;
    RcPlan plan;
    RcBuf kern;
    RcBuf I[RC_MAX_LEVELS], RA[RC_MAX_LEVELS], RB[RC_MAX_LEVELS];
    RcBuf FA[RC_MAX_LEVELS], FB[RC_MAX_LEVELS];
    RcBuf stage_u8, stage_flow, stage_f32[4];
    RcBuf exM, exV, exG;       // option "exact": matrix planes, window column sums, window row sums (box)
    // host-pointer frame loop (rcflow_push_frame_u8): two page-locked staging frames, an event per frame that
    // fires when its upload has left the staging buffer, two device frames (stage_u8) and the resident flow
    void* pin[2] = {nullptr, nullptr};
    size_t pin_bytes = 0;
    hipEvent_t pin_free[2] = {nullptr, nullptr};
    int pin_i = 0;
    int pin_acq = -1, pin_w = 0, pin_h = 0;   // staging buffer handed to the host by rcflow_frame_buffer_acquire, not yet pushed
    int flow_w = 0, flow_h = 0;   // size of the flow field resident in stage_flow (0: none yet)
    RcBuf lk;                  // sparse PyrLK pyramids + derivatives (lk_kernels.hip)
    RcBuf area_tab;            // INTER_AREA decimation tables
    int primed = 0, cur_slot = 0;
    // lockstep batch of streams (rcflow_push_batch_dev): parity of the ring, captured graphs
    int batch_primed = 0, batch_cur = 0;
    void* batch_exec[2] = {nullptr, nullptr};
    int batch_eager[2] = {0, 0};
    RcBatchKey batch_key[2] = {};
    // rcflow_frame_loop_step: one captured launch sequence per ring parity, the configuration they were captured for,
    // the stream's flow-field counter (host copy; the kernels read the device word an.loopc)
    void* loop_exec[2] = {nullptr, nullptr};
    int loop_eager[2] = {0, 0};
    unsigned char loop_key[2][160] = {};
    int loop_fc = 0;
    RcAnalysis an;
};

struct RcProfRec {
    int id;
    hipEvent_t e0, e1;
    double bytes;         // compulsory bytes of this launch (inputs once + outputs once)
    double model_bytes;   // SURVEY.md section 8(d) algorithmic bytes of the stages it covers
};

struct rc_ctx {
    int device = 0;
    int max_w = 0, max_h = 0, nstreams = 0;
    RcSlot* slots = nullptr;
    int chunk = 32;
    int exact_taps = 0;
    int exact = -1;            // option "exact": upstream's CPU operation order (exact_kernels.hip); -1 = where the fast path cannot hold the tolerance
    int fuse_iters = 1;
    int chain_min_blocks = 0;  // option "chain_min_blocks" (0 = the kernel's default, 4096)
    int chain = 8;             // option "chain": pairs per tile chain of the fused winsize-3 flow kernel (1 = off)
    int xcd_remap = 1;
    int poly_tile_h = 32;
    int frame_overlap = 1;     // option "frame_overlap": frame loop with the expansion of frame t+1 on a second stream beside the flow kernels of frame t
    int merge_small = 1;       // option "merge_small": merged pyramid / expansion launches for calls of one or two frames
    int overlap = 0;           // option "overlap": clip path on two streams (measured: no gain, the grids fill the GPU)
    int poly_mfma = 0;          // option "poly_mfma": vertical pass of the expansion on the matrix cores (measured 25 % slower)
    int hist_blocks = 0;       // option "hist_blocks": cap on histogram blocks per launch (0 = default)
    int fuse_pyr = 1;         // scale-0 expansion also writes pyramid scales 1 and 2 (exact 2:1 / 4:1 sizes)
    int ablate = 0;
    void* comm = nullptr;      // RcComm (comm_rccl.hip): the histogram all-reduce over RCCL
    void* stamps = nullptr;
    int prof_on = 0;
    std::vector<RcProfRec> prof_pending;
    std::vector<hipEvent_t> ev_pool;
    std::vector<int> prof_launches;
    std::vector<double> prof_ms, prof_bytes, prof_model_bytes;
};

enum { RC_K_PYR = 0, RC_K_POLY = 1, RC_K_ITER = 2, RC_K_HIST = 3, RC_K_THRESH = 4, RC_K_CLASSIFY = 5,
       RC_K_ADVECT_FIELD = 6, RC_K_ADVECT_POINTS = 7, RC_K_POSTOP = 8, RC_K_COLOR = 9, RC_K_ITER2 = 10,
       RC_K_PREPROC = 11, RC_K_EDGES = 12, RC_K_DISPLAY = 13, RC_K_HSV2BGR = 14, RC_K_OVERLAY = 15, RC_K_KINDS = 16 };

void rc_set_error(const char* fmt, ...);
int rc_buf_ensure(RcBuf& b, size_t bytes);
void rc_buf_free(RcBuf& b);
RcSlot* rc_slot(rc_ctx* ctx, int stream);
void rc_batch_graph_drop(RcSlot& s);
void rc_loop_graph_drop(RcSlot& s);
// analysis_kernels.hip, for rcflow_frame_loop_step
int rc_classify_accumulate(rc_ctx* ctx, int stream, const float* d_flow, size_t flow_step, int w, int h, int framecount,
                           float MID, float LOWER, float* d_polar, size_t polar_step, float* d_wclass, size_t wc_step,
                           float* d_out, size_t out_step, uint8_t* d_mask, size_t mask_step);
int rc_loop_counter(rc_ctx* ctx, RcSlot& s, bool set, int value);
int rc_hist_book(RcSlot& s, int w, int h, bool commit);
int rc_analysis_ensure(rc_ctx* ctx, RcSlot& s, int w, int h);

struct RcProfScope {
    rc_ctx* ctx;
    hipStream_t s;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    int id;
    double bytes, model_bytes;
    RcProfScope(rc_ctx* c, hipStream_t st, int kind, int level, double alg_bytes, double survey_bytes = -1.);
    ~RcProfScope();
};

#define RC_HIP(call)                                                                  \
    do {                                                                              \
        hipError_t e_ = (call);                                                       \
        if (e_ != hipSuccess) {                                                       \
            rc_set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
            return RC_EHIP;                                                           \
        }                                                                             \
    } while (0)
