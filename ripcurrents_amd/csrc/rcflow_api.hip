// rcflow_api.hip -- C ABI of librcflow.so: context, plans, level driver (A7) and the
// Farneback entry points.  See include/rcflow.h for the reference interfaces replaced.

#include <cfloat>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>

#include "rc_host.h"

// ---------------------------------------------------------------------------- errors
static thread_local char g_err[512] = "";
void rc_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
extern "C" const char* rcflow_last_error(void) { return g_err; }
extern "C" int rcflow_abi_version(void) { return RCFLOW_ABI_VERSION; }

int rc_buf_ensure(RcBuf& b, size_t bytes) {
    if (b.bytes >= bytes && b.p) return RC_OK;
    if (b.p) (void)hipFree(b.p);
    b.p = nullptr;
    b.bytes = 0;
    if (bytes == 0) return RC_OK;
    hipError_t e = hipMalloc(&b.p, bytes);
    if (e != hipSuccess) {
        rc_set_error("hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
        b.p = nullptr;
        return RC_ENOMEM;
    }
    b.bytes = bytes;
    return RC_OK;
}
void rc_buf_free(RcBuf& b) {
    if (b.p) (void)hipFree(b.p);
    b.p = nullptr;
    b.bytes = 0;
}
RcSlot* rc_slot(rc_ctx* ctx, int stream) {
    if (!ctx || stream < 0 || stream >= ctx->nstreams) {
        rc_set_error("bad context or stream index %d", stream);
        return nullptr;
    }
    return &ctx->slots[stream];
}

// ---------------------------------------------------------------------------- profiling
RcProfScope::RcProfScope(rc_ctx* c, hipStream_t st, int kind, int level, double alg_bytes, double survey_bytes)
    : ctx(c), s(st), id(kind * RC_MAX_LEVELS + level), bytes(alg_bytes),
      model_bytes(survey_bytes < 0 ? alg_bytes : survey_bytes) {
    if (!ctx->prof_on) return;
    for (int i = 0; i < 2; i++) {
        hipEvent_t e;
        if (!ctx->ev_pool.empty()) {
            e = ctx->ev_pool.back();
            ctx->ev_pool.pop_back();
        } else if (hipEventCreate(&e) != hipSuccess) {
            e = nullptr;
        }
        (i ? e1 : e0) = e;
    }
    if (e0) (void)hipEventRecord(e0, s);
}
RcProfScope::~RcProfScope() {
    if (!ctx->prof_on || !e0 || !e1) return;
    (void)hipEventRecord(e1, s);
    ctx->prof_pending.push_back({id, e0, e1, bytes, model_bytes});
}

static const char* kKindNames[RC_K_KINDS] = {"pyr_level", "polyexp", "flow_iter", "polar_hist",
                                             "thresholds", "classify_accumulate", "advect_field",
                                             "advect_points", "flow_postop", "flow_color", "flow_iter_x2",
                                             "frame_preproc", "create_edges", "streamline_display", "hsv_to_bgr",
                                             "create_output"};
// The reference's wall-clock buckets (ripcurrents.cpp:103-109, sampled at :205,223,293,314,411,483, printed at
// :518-524) and the kernels that do each bucket's work here.  time_polar has no kernel of its own: the
// cartToPolar of :305-309 is fused into the histogram and classification kernels; classify_accumulate spans
// :376-439 (the reference samples time_threshold at :411, inside it) and is booked under "threshold";
// time_codec (video decode) is host I/O outside this library.
static const int kBucketOfKind[RC_K_KINDS] = {
    /* pyr_level */ 0, /* polyexp */ 0, /* flow_iter */ 0, /* polar_hist */ 2, /* thresholds */ 2,
    /* classify_accumulate */ 2, /* advect_field */ 6, /* advect_points */ 6, /* flow_postop */ 0, /* flow_color */ 2,
    /* flow_iter_x2 */ 0, /* frame_preproc */ 0, /* create_edges */ 4, /* streamline_display */ 6, /* hsv_to_bgr */ 2,
    /* create_output */ 3};
static const char* kBucketNames[RC_PROFILE_BUCKETS] = {"farneback", "polar", "threshold", "overlay", "erosion", "codec", "stream"};
static char g_names[RC_K_KINDS * RC_MAX_LEVELS][40];

static void prof_resolve(rc_ctx* ctx) {
    size_t n = RC_K_KINDS * RC_MAX_LEVELS;
    if (ctx->prof_ms.size() != n) {
        ctx->prof_ms.assign(n, 0.);
        ctx->prof_bytes.assign(n, 0.);
        ctx->prof_model_bytes.assign(n, 0.);
        ctx->prof_launches.assign(n, 0);
    }
    for (auto& r : ctx->prof_pending) {
        float ms = 0.f;
        (void)hipEventSynchronize(r.e1);
        if (hipEventElapsedTime(&ms, r.e0, r.e1) == hipSuccess) {
            ctx->prof_ms[r.id] += ms;
            ctx->prof_bytes[r.id] += r.bytes;
            ctx->prof_model_bytes[r.id] += r.model_bytes;
            ctx->prof_launches[r.id]++;
        }
        ctx->ev_pool.push_back(r.e0);
        ctx->ev_pool.push_back(r.e1);
    }
    ctx->prof_pending.clear();
}

extern "C" int rcflow_profile_enable(rc_ctx* ctx, int on) {
    if (!ctx) return RC_EINVAL;
    if (!on) prof_resolve(ctx);
    ctx->prof_on = on ? 1 : 0;
    return RC_OK;
}
extern "C" int rcflow_profile_reset(rc_ctx* ctx) {
    if (!ctx) return RC_EINVAL;
    prof_resolve(ctx);
    ctx->prof_ms.assign(ctx->prof_ms.size(), 0.);
    ctx->prof_bytes.assign(ctx->prof_bytes.size(), 0.);
    ctx->prof_model_bytes.assign(ctx->prof_model_bytes.size(), 0.);
    ctx->prof_launches.assign(ctx->prof_launches.size(), 0);
    return RC_OK;
}
extern "C" int rcflow_profile_read(rc_ctx* ctx, int cap, const char** names, int* launches,
                                   double* total_ms, double* alg_bytes, double* model_bytes) {
    if (!ctx) return RC_EINVAL;
    prof_resolve(ctx);
    int n = 0;
    for (size_t id = 0; id < ctx->prof_ms.size() && n < cap; id++) {
        if (!ctx->prof_launches[id]) continue;
        snprintf(g_names[id], sizeof(g_names[id]), "%s@%d", kKindNames[id / RC_MAX_LEVELS],
                 (int)(id % RC_MAX_LEVELS));
        if (names) names[n] = g_names[id];
        if (launches) launches[n] = ctx->prof_launches[id];
        if (total_ms) total_ms[n] = ctx->prof_ms[id];
        if (alg_bytes) alg_bytes[n] = ctx->prof_bytes[id];
        if (model_bytes) model_bytes[n] = ctx->prof_model_bytes[id];
        n++;
    }
    return n;
}

extern "C" int rcflow_profile_read_buckets(rc_ctx* ctx, const char** names, double* ms) {
    if (!ctx) return RC_EINVAL;
    prof_resolve(ctx);
    double b[RC_PROFILE_BUCKETS] = {0};
    for (size_t id = 0; id < ctx->prof_ms.size(); id++) b[kBucketOfKind[id / RC_MAX_LEVELS]] += ctx->prof_ms[id];
    for (int i = 0; i < RC_PROFILE_BUCKETS; i++) {
        if (names) names[i] = kBucketNames[i];
        if (ms) ms[i] = b[i];
    }
    return RC_PROFILE_BUCKETS;
}

// ---------------------------------------------------------------------------- lifetime
extern "C" int rcflow_create(rc_ctx** out, int device, int max_w, int max_h, int max_streams) {
    if (!out || max_w <= 0 || max_h <= 0 || max_streams <= 0 || max_streams > 256) {
        rc_set_error("rcflow_create: bad arguments");
        return RC_EINVAL;
    }
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device < 0 || device >= ndev) {
        rc_set_error("rcflow_create: no usable HIP device (count=%d, requested %d)", ndev, device);
        return RC_ENODEV;
    }
    RC_HIP(hipSetDevice(device));
    hipDeviceProp_t prop;
    RC_HIP(hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        rc_set_error("rcflow_create: device %d is %s; this library holds gfx950 code only", device,
                     prop.gcnArchName);
        return RC_ENODEV;
    }
    rc_ctx* ctx = new rc_ctx();
    ctx->device = device;
    ctx->max_w = max_w;
    ctx->max_h = max_h;
    ctx->nstreams = max_streams;
    ctx->slots = new RcSlot[max_streams];
    for (int i = 0; i < max_streams; i++) {
        hipError_t e = hipStreamCreateWithFlags(&ctx->slots[i].own, hipStreamNonBlocking);
        if (e != hipSuccess) {
            rc_set_error("hipStreamCreate failed: %s", hipGetErrorString(e));
            rcflow_destroy(ctx);
            return RC_EHIP;
        }
        ctx->slots[i].cur = ctx->slots[i].own;
    }
    *out = ctx;
    return RC_OK;
}

static void slot_free(RcSlot& s) {
    rc_buf_free(s.kern);
    for (int k = 0; k < RC_MAX_LEVELS; k++) {
        rc_buf_free(s.I[k]); rc_buf_free(s.RA[k]); rc_buf_free(s.RB[k]);
        rc_buf_free(s.FA[k]); rc_buf_free(s.FB[k]);
    }
    rc_batch_graph_drop(s);
    rc_buf_free(s.stage_u8); rc_buf_free(s.stage_flow); rc_buf_free(s.lk); rc_buf_free(s.area_tab);
    rc_buf_free(s.exM); rc_buf_free(s.exV); rc_buf_free(s.exG);
    for (int i = 0; i < 2; i++) {
        if (s.pin[i]) (void)hipHostFree(s.pin[i]);
        if (s.pin_free[i]) (void)hipEventDestroy(s.pin_free[i]);
        s.pin[i] = nullptr; s.pin_free[i] = nullptr;
    }
    s.pin_bytes = 0;
    for (auto& b : s.stage_f32) rc_buf_free(b);
    rc_buf_free(s.an.hist); rc_buf_free(s.an.hist_part); rc_buf_free(s.an.thr); rc_buf_free(s.an.acc);
    rc_buf_free(s.an.pt); rc_buf_free(s.an.dist); rc_buf_free(s.an.scratch); rc_buf_free(s.an.jet); rc_buf_free(s.an.loopc);
    rc_loop_graph_drop(s);
    for (auto& e : s.fev) { if (e) (void)hipEventDestroy(e); e = nullptr; }
    for (auto& e : s.flow_done) { if (e) (void)hipEventDestroy(e); e = nullptr; }
    if (s.own) (void)hipStreamDestroy(s.own);
    if (s.aux) (void)hipStreamDestroy(s.aux);
    s.aux = nullptr;
    s.own = s.cur = nullptr;
}

extern "C" void rcflow_destroy(rc_ctx* ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    (void)hipDeviceSynchronize();
    (void)rcflow_comm_destroy(ctx);
    prof_resolve(ctx);
    for (auto e : ctx->ev_pool) (void)hipEventDestroy(e);
    if (ctx->slots) {
        for (int i = 0; i < ctx->nstreams; i++) slot_free(ctx->slots[i]);
        delete[] ctx->slots;
    }
    delete ctx;
}

extern "C" int rcflow_sync(rc_ctx* ctx, int stream) {
    RcSlot* s = rc_slot(ctx, stream);
    if (!s) return RC_EINVAL;
    RC_HIP(hipStreamSynchronize(s->cur));
    return RC_OK;
}

extern "C" int rcflow_set_hip_stream(rc_ctx* ctx, int stream, void* hip_stream) {
    RcSlot* s = rc_slot(ctx, stream);
    if (!s) return RC_EINVAL;
    s->cur = (hipStream_t)hip_stream;
    return RC_OK;
}

extern "C" int rcflow_use_own_stream(rc_ctx* ctx, int stream) {
    RcSlot* s = rc_slot(ctx, stream);
    if (!s) return RC_EINVAL;
    s->cur = s->own;
    return RC_OK;
}

extern "C" int rcflow_debug_read_stamps(rc_ctx* ctx, long long* out, int n) {
    if (!ctx || !ctx->stamps || !out || n > 8 * 4096) return RC_EINVAL;
    RC_HIP(hipDeviceSynchronize());
    RC_HIP(hipMemcpy(out, ctx->stamps, (size_t)n * 8, hipMemcpyDeviceToHost));
    return RC_OK;
}

// Diagnostic (not part of include/rcflow.h; tests/test_cabi_and_host.py): the pair groups the fused winsize-3 kernel would
// walk for a launch of `pairs` consecutive pairs of a w x h scale with option chain = `chain` -- pure host logic, no GPU.
extern "C" int rcflow_debug_chain_plan(int w, int h, int pairs, int chain, int force, int* starts, int cap) {
    if (w <= 0 || h <= 0 || pairs < 1 || !starts || cap < 2) return RC_EINVAL;
    RcIterArgs a;
    memset(&a, 0, sizeof(a));
    a.w = w; a.h = h; a.chain = chain; a.addr32 = 1; a.slot0 = 0; a.slot1 = 1; a.zstep = 1; a.nslots = pairs + 1;
    a.ablate = force ? RC_ABL_FORCE_CHAIN : 0;
    return rc_flow_fast::rc_flow_chain_groups(a, pairs, starts, cap);
}

// Diagnostic (not part of include/rcflow.h; scripts/r3/level_errors.py): the flow field a scale's last launch wrote
// for pair 0 of the slot's last call -- valid for scales >= 1 after a call whose iterations fit one or two launches
// per scale (buffer A then B alternate); which = 0 / 1 picks the buffer.
extern "C" int rcflow_debug_level_flow_ptr(rc_ctx* ctx, int stream, int level, int which, float** d_flow, int* w, int* h) {
    RcSlot* s = rc_slot(ctx, stream);
    if (!s || !s->plan.valid || level < 0 || level >= s->plan.nlev || !d_flow) return RC_EINVAL;
    *d_flow = (float*)(which ? s->FB[level].p : s->FA[level].p);
    if (w) *w = s->plan.lv[level].w;
    if (h) *h = s->plan.lv[level].h;
    return *d_flow ? RC_OK : RC_ESTATE;
}

extern "C" int rcflow_set_option(rc_ctx* ctx, const char* name, int value) {
    if (!ctx || !name) return RC_EINVAL;
    if (!strcmp(name, "chunk")) {
        if (value < 1 || value > 64) return RC_EINVAL;
        ctx->chunk = value;
    } else if (!strcmp(name, "exact_taps")) {
        ctx->exact_taps = value ? 1 : 0;
    } else if (!strcmp(name, "exact")) {
        if (value < -1 || value > 1) return RC_EINVAL;
        ctx->exact = value;
    } else if (!strcmp(name, "fuse_iters")) {
        ctx->fuse_iters = value ? 1 : 0;
    } else if (!strcmp(name, "chain")) {
        if (value < 1 || value > 64) return RC_EINVAL;
        ctx->chain = value;
    } else if (!strcmp(name, "chain_min_blocks")) {
        if (value < 0) return RC_EINVAL;
        ctx->chain_min_blocks = value;
    } else if (!strcmp(name, "xcd_remap")) {
        ctx->xcd_remap = value ? 1 : 0;
    } else if (!strcmp(name, "hist_blocks")) {
        if (value < 0 || value > 65535) return RC_EINVAL;
        ctx->hist_blocks = value;
    } else if (!strcmp(name, "overlap")) {
        ctx->overlap = value ? 1 : 0;
    } else if (!strcmp(name, "frame_overlap")) {
        if (value < 0 || value > 2) return RC_EINVAL;
        ctx->frame_overlap = value;
    } else if (!strcmp(name, "merge_small")) {
        ctx->merge_small = value ? 1 : 0;
    } else if (!strcmp(name, "poly_mfma")) {
        ctx->poly_mfma = value ? 1 : 0;
    } else if (!strcmp(name, "poly_tile_h")) {
        if (value != 32 && value != 48) return RC_EINVAL;
        ctx->poly_tile_h = value;
    } else if (!strcmp(name, "fuse_pyr")) {
        ctx->fuse_pyr = value != 0;
    } else if (!strcmp(name, "ablate")) {
        ctx->ablate = value;
    } else if (!strcmp(name, "stamps")) {
        // diagnostic: value != 0 allocates a stamp buffer that the scale-0 flow kernel fills
        if (value && !ctx->stamps) {
            if (hipMalloc(&ctx->stamps, 8 * 8 * 4096) != hipSuccess) return RC_ENOMEM;
            (void)hipMemset(ctx->stamps, 0, 8 * 8 * 4096);
        } else if (!value && ctx->stamps) {
            (void)hipFree(ctx->stamps);
            ctx->stamps = nullptr;
        }
    } else {
        rc_set_error("unknown option %s", name);
        return RC_EINVAL;
    }
    for (int i = 0; i < ctx->nstreams; i++) ctx->slots[i].plan.valid = false;
    return RC_OK;
}

// ---------------------------------------------------------------------------- plan (A7 geometry)
static inline int cv_round(double v) { return (int)nearbyint(v); }   // round half to even

static int crop_levels(int w, int h, double pyr_scale, int levels) {
    const int min_size = 32;   // optflow.cpp calc()
    int k;
    double scale = 1;
    for (k = 0; k < levels; k++) {
        scale *= pyr_scale;
        if (w * scale < min_size || h * scale < min_size) break;
    }
    return k;
}

static void level_geom(int w, int h, double pyr_scale, int k, RcLevel& L) {
    double scale = 1;
    for (int i = 0; i < k; i++) scale *= pyr_scale;
    L.sigma = (1. / scale - 1) * 0.5;
    int smooth_sz = cv_round(L.sigma * 5) | 1;
    L.ksize = smooth_sz > 3 ? smooth_sz : 3;
    L.w = cv_round(w * scale);
    L.h = cv_round(h * scale);
    L.scale_x = 1. / ((double)L.w / w);
    L.scale_y = 1. / ((double)L.h / h);
}

extern "C" int rcflow_level_geometry(int w, int h, double pyr_scale, int levels, int k, int* wk, int* hk) {
    if (w <= 0 || h <= 0 || !(pyr_scale > 0 && pyr_scale < 1) || levels < 0 || k < 0) return RC_EINVAL;
    RcLevel L;
    level_geom(w, h, pyr_scale, k, L);
    if (wk) *wk = L.w;
    if (hk) *hk = L.h;
    return crop_levels(w, h, pyr_scale, levels);
}

// smooth.cpp getGaussianKernel(n, sigma, CV_32F)
static void host_gaussian_kernel(int n, double sigma, float* cf) {
    static const float tab1[] = {1.f};
    static const float tab3[] = {0.25f, 0.5f, 0.25f};
    static const float tab5[] = {0.0625f, 0.25f, 0.375f, 0.25f, 0.0625f};
    static const float tab7[] = {0.03125f, 0.109375f, 0.21875f, 0.28125f, 0.21875f, 0.109375f, 0.03125f};
    const float* fixed = nullptr;
    if (n % 2 == 1 && n <= 7 && sigma <= 0) fixed = n == 1 ? tab1 : n == 3 ? tab3 : n == 5 ? tab5 : tab7;
    double sx = sigma > 0 ? sigma : ((n - 1) * 0.5 - 1) * 0.3 + 0.8;
    double scale2x = -0.5 / (sx * sx), sum = 0;
    for (int i = 0; i < n; i++) {
        double x = i - (n - 1) * 0.5;
        double t = fixed ? (double)fixed[i] : exp(scale2x * x * x);
        cf[i] = (float)t;
        sum += cf[i];
    }
    sum = 1. / sum;
    for (int i = 0; i < n; i++) cf[i] = (float)(cf[i] * sum);
}

// optflow.cpp FarnebackPrepareGaussian; the 6x6 moment matrix is inverted by Cholesky.
static int host_prepare_poly(int n, double sigma, int exact_taps, RcPolyK& pk) {
    if (sigma < FLT_EPSILON) sigma = n * 0.3;
    std::vector<float> gb(2 * n + 1), xgb(2 * n + 1), xxgb(2 * n + 1);
    float *g = gb.data() + n, *xg = xgb.data() + n, *xxg = xxgb.data() + n;
    double s = 0.;
    for (int x = -n; x <= n; x++) {
        g[x] = (float)exp(-x * x / (2 * sigma * sigma));
        s += g[x];
    }
    s = 1. / s;
    for (int x = -n; x <= n; x++) {
        g[x] = (float)(g[x] * s);
        xg[x] = (float)(x * g[x]);
        xxg[x] = (float)(x * x * g[x]);
    }
    double G[6][6] = {{0}};
    for (int y = -n; y <= n; y++)
        for (int x = -n; x <= n; x++) {
            G[0][0] += g[y] * g[x];
            G[1][1] += g[y] * g[x] * x * x;
            G[3][3] += g[y] * g[x] * x * x * x * x;
            G[5][5] += g[y] * g[x] * x * x * y * y;
        }
    G[2][2] = G[0][3] = G[0][4] = G[3][0] = G[4][0] = G[1][1];
    G[4][4] = G[3][3];
    G[3][4] = G[4][3] = G[5][5];
    // invG = G.inv(DECOMP_CHOLESKY): cv::invert -> hal::Cholesky64f on the identity (core/src/matrix_decomp.cpp
    // CholImpl<double>: 1/sqrt(pivot) on the diagonal, forward then backward substitution), restated
    // operation for operation so that the four scalars carry upstream's bits.
    double L[6][6], inv[6][6];
    for (int i = 0; i < 6; i++)
        for (int j = 0; j < 6; j++) { L[i][j] = G[i][j]; inv[i][j] = i == j ? 1. : 0.; }
    for (int i = 0; i < 6; i++) {
        double v;
        int j, k;
        for (j = 0; j < i; j++) {
            v = L[i][j];
            for (k = 0; k < j; k++) v -= L[i][k] * L[j][k];
            L[i][j] = v * L[j][j];
        }
        v = L[i][i];
        for (k = 0; k < j; k++) { double t = L[i][k]; v -= t * t; }
        if (!(v >= DBL_EPSILON)) return RC_EINVAL;
        L[i][i] = 1. / sqrt(v);
    }
    for (int i = 0; i < 6; i++)
        for (int j = 0; j < 6; j++) {
            double v = inv[i][j];
            for (int k = 0; k < i; k++) v -= L[i][k] * inv[k][j];
            inv[i][j] = v * L[i][i];
        }
    for (int i = 5; i >= 0; i--)
        for (int j = 0; j < 6; j++) {
            double v = inv[i][j];
            for (int k = 5; k > i; k--) v -= L[k][i] * inv[k][j];
            inv[i][j] = v * L[i][i];
        }
    pk.ig11 = inv[1][1];
    pk.ig03 = inv[0][3];
    pk.ig33 = inv[3][3];
    pk.ig55 = inv[5][5];
    pk.n = n;
    // Taps whose combined weight cannot change any sum beyond 1e-9 of its kernel mass are
    // dropped (poly_n = 15 with sigma = 1.2 evaluates 19 of its 31 taps; "exact_taps"
    // keeps all of them).
    int n_thr = n;
    if (!exact_taps) {
        double m0 = 0, m1 = 0, m2 = 0;
        for (int k = 1; k <= n; k++) { m0 += g[k]; m1 += fabs(xg[k]); m2 += xxg[k]; }
        double t0 = 0, t1 = 0, t2 = 0;
        for (int k = n; k >= 1; k--) {
            t0 += g[k]; t1 += fabs(xg[k]); t2 += xxg[k];
            if (t0 > 1e-8 * (m0 + g[0]) || t1 > 1e-8 * m1 || t2 > 1e-8 * m2) break;
            n_thr = k - 1;
        }
        if (n_thr < 1) n_thr = 1;
    }
    static const int inst[] = {3, 5, 7, 8, 9, 12, 16, 24, 32};
    int R = 32;
    for (int v : inst)
        if (v >= n_thr) { R = v; break; }
    pk.n_eff = R < n ? R : n;
    memset(pk.g, 0, sizeof(pk.g));
    memset(pk.xg, 0, sizeof(pk.xg));
    memset(pk.xxg, 0, sizeof(pk.xxg));
    double sg = 0, s2 = 0;
    for (int k = 0; k <= pk.n_eff; k++) {
        pk.g[k] = g[k];
        pk.xg[k] = xg[k];
        pk.xxg[k] = xxg[k];
        sg += (k ? 2. : 1.) * g[k];
        s2 += (k ? 2. : 0.) * xxg[k];
    }
    pk.kdc = sg * sg * pk.ig03 + sg * s2 * pk.ig33;
    return RC_OK;
}

static void host_window(int winsize, int flags, RcWindow& win) {
    int m = winsize / 2;
    memset(&win, 0, sizeof(win));
    win.m = m;
    win.gaussian = (flags & RC_FARNEBACK_GAUSSIAN) ? 1 : 0;
    win.box_scale = 1. / ((double)winsize * winsize);
    win.box_eps = 1e-3 / (win.box_scale * win.box_scale);
    double sigma = m * 0.3, s = 1;
    win.k[0] = (float)s;
    for (int i = 1; i <= m; i++) {
        float t = (float)exp(-i * i / (2 * sigma * sigma));
        win.k[i] = t;
        s += t * 2;
    }
    s = 1. / s;
    for (int i = 0; i <= m; i++) win.k[i] = (float)(win.k[i] * s);
}

static void pick_pyr_tile(RcLevel& L, int W0, int H0) {
    // tw <= 128 and th <= 128 (the coordinate tables are filled by threads 0..127 / 128..255)
    static const int tiles[][2] = {{64, 16}, {64, 8}, {64, 4}, {32, 8}, {16, 8}, {16, 4}, {8, 4}, {4, 4}, {2, 2}, {1, 1}};
    int r = L.ksize / 2;
    for (auto& t : tiles) {
        int tw = t[0], th = t[1];
        int rw = (int)ceil(tw * L.scale_x) + 2 * r + 4;
        int rh = (int)ceil(th * L.scale_y) + 2 * r + 4;
        if (rw > W0 + 2 * r + 2) rw = W0 + 2 * r + 2;
        if (rh > H0 + 2 * r + 2) rh = H0 + 2 * r + 2;
        int rwp = (rw + 15) & ~15;
        size_t lds = (size_t)rh * rwp + sizeof(float) * ((size_t)rh * 2 * tw + 3 * tw + 3 * th + L.ksize);
        if (lds <= 40 * 1024 || tw == 1) {
            L.pyr_tw = tw; L.pyr_th = th; L.pyr_reg_w = rwp; L.pyr_reg_h = rh; L.pyr_lds = lds;
            return;
        }
    }
}

static int params_valid(const rc_farneback_params* p) {
    if (!p) return 0;
    if (!(p->pyr_scale > 0 && p->pyr_scale < 1)) return 0;
    if (p->levels < 0 || p->levels >= RC_MAX_LEVELS) return 0;
    if (p->winsize < 1 || p->winsize / 2 > 24) return 0;
    if (p->iterations < 0 || p->iterations > 1000) return 0;
    if (p->poly_n < 1 || p->poly_n > RC_MAX_POLY_N) return 0;
    if (!(p->poly_sigma >= 0)) return 0;
    if (p->flags & ~RC_FARNEBACK_GAUSSIAN) return 0;   // USE_INITIAL_FLOW unsupported
    return 1;
}

static int ensure_plan(rc_ctx* ctx, RcSlot& s, int w, int h, const rc_farneback_params* p, int chunk, int nslots = 0) {
    if (nslots <= 0) nslots = chunk + 1;
    if (w <= 0 || h <= 0 || !params_valid(p)) {
        rc_set_error("invalid Farneback arguments (w=%d h=%d)", w, h);
        return RC_EINVAL;
    }
    if (w > ctx->max_w || h > ctx->max_h) {
        rc_set_error("frame %dx%d exceeds the context's %dx%d", w, h, ctx->max_w, ctx->max_h);
        return RC_ESIZE;
    }
    RcPlan& pl = s.plan;
    // field by field: the struct has tail padding that a caller's brace-initialised copy leaves indeterminate
    const rc_farneback_params& q = pl.prm;
    const bool same_prm = q.pyr_scale == p->pyr_scale && q.levels == p->levels && q.winsize == p->winsize &&
                          q.iterations == p->iterations && q.poly_n == p->poly_n && q.poly_sigma == p->poly_sigma &&
                          q.flags == p->flags;
    // Option "exact" = -1 (default): upstream's operation order wherever the window is a near-pointwise
    // solve (Gaussian winsize < 7: sigma = 0.3 m <= 0.6, the centre tap carries >= half of the weight;
    // winsize 1) -- there the 2x2 determinant vanishes on smooth regions and the fast kernels' rounding
    // differences are amplified beyond any tolerance (main.cpp:264).
    const int exact = ctx->exact >= 0 ? ctx->exact
                                      : (p->winsize / 2 == 0 || ((p->flags & RC_FARNEBACK_GAUSSIAN) && p->winsize / 2 <= 2));
    if (pl.valid && pl.w == w && pl.h == h && same_prm && pl.chunk == chunk &&
        pl.nslots == nslots && pl.exact_taps == ctx->exact_taps && pl.exact == exact)
        return RC_OK;
    rc_batch_graph_drop(s);
    rc_loop_graph_drop(s);
    RC_HIP(hipStreamSynchronize(s.cur));
    if (s.aux) RC_HIP(hipStreamSynchronize(s.aux));
    pl.valid = false;
    pl.w = w; pl.h = h; pl.chunk = chunk; pl.nslots = nslots;
    memset(&pl.prm, 0, sizeof(pl.prm));
    pl.prm.pyr_scale = p->pyr_scale; pl.prm.levels = p->levels; pl.prm.winsize = p->winsize;
    pl.prm.iterations = p->iterations; pl.prm.poly_n = p->poly_n; pl.prm.poly_sigma = p->poly_sigma;
    pl.prm.flags = p->flags;
    pl.exact_taps = ctx->exact_taps;
    pl.exact = exact;
    int L = crop_levels(w, h, p->pyr_scale, p->levels);
    pl.nlev = L + 1;
    size_t kern_total = 0;
    for (int k = 0; k <= L; k++) {
        level_geom(w, h, p->pyr_scale, k, pl.lv[k]);
        if (pl.lv[k].ksize > 1023) { rc_set_error("pyramid blur too wide"); return RC_EINVAL; }
        pick_pyr_tile(pl.lv[k], w, h);
        pl.kern_off[k] = kern_total;
        kern_total += (pl.lv[k].ksize + 3) & ~3;
    }
    int rc = host_prepare_poly(p->poly_n, p->poly_sigma, ctx->exact_taps || exact, pl.pk);
    if (rc) { rc_set_error("polynomial-expansion moment matrix is not positive definite"); return rc; }
    host_window(p->winsize, p->flags, pl.win);

    std::vector<float> kh(kern_total, 0.f);
    for (int k = 0; k <= L; k++) host_gaussian_kernel(pl.lv[k].ksize, pl.lv[k].sigma > 0 ? pl.lv[k].sigma : 0., kh.data() + pl.kern_off[k]);
    if ((rc = rc_buf_ensure(s.kern, kern_total * sizeof(float)))) return rc;
    RC_HIP(hipMemcpy(s.kern.p, kh.data(), kern_total * sizeof(float), hipMemcpyHostToDevice));
    for (int k = 0; k <= L; k++) {
        size_t n = (size_t)pl.lv[k].w * pl.lv[k].h;
        if ((k > 0 || exact) && (rc = rc_buf_ensure(s.I[k], n * pl.nslots * sizeof(float)))) return rc;
        if ((rc = rc_buf_ensure(s.RA[k], n * pl.nslots * sizeof(float4)))) return rc;
        if ((rc = rc_buf_ensure(s.RB[k], n * pl.nslots * sizeof(float)))) return rc;
        if ((rc = rc_buf_ensure(s.FA[k], n * chunk * sizeof(float2)))) return rc;
        if ((rc = rc_buf_ensure(s.FB[k], n * chunk * sizeof(float2)))) return rc;
    }
    s.primed = 0;
    s.batch_primed = 0;
    pl.valid = true;
    return RC_OK;
}

// ---------------------------------------------------------------------------- level driver
// Pyramid + polynomial expansion of `count` frames into R slots dslot0.. (A1 + A2).
static int expand_frames(rc_ctx* ctx, RcSlot& s, const uint8_t* d_src, size_t frame_stride, size_t step,
                         int count, int dslot0, int zstep = 1) {
    if (!s.in_ts_push) s.ts_streak = 0;      // work on the slot outside a two-stream frame push (see RcSlot::ts_streak)
    RcPlan& pl = s.plan;
    RcPolyArgs qa[RC_MAX_LEVELS];
    RcPyrArgs pa[RC_MAX_LEVELS];
    for (int k = 0; k < pl.nlev; k++) {
        const RcLevel& L = pl.lv[k];
        size_t n = (size_t)L.w * L.h;
        RcPolyArgs& q = qa[k];
        memset(&q, 0, sizeof(q));
        q.RA = (float4*)s.RA[k].p; q.RB = (float*)s.RB[k].p; q.R_slot_stride = n;
        q.slot0 = dslot0; q.nslots = pl.nslots; q.zstep = zstep; q.w = L.w; q.h = L.h; q.pk = pl.pk;
        q.tile_h = ctx->poly_tile_h; q.no_fast_u8 = (ctx->ablate & RC_ABL_NO_FAST_U8) != 0; q.valu_vertical = !ctx->poly_mfma;
        q.stamps = (k == 0) ? (long long*)ctx->stamps : nullptr;
        if (k == 0 && !pl.exact) {
            // scale 0: pyramid (3x3 blur, identity resize) fused into the expansion
            q.src8 = d_src; q.src8_step = step; q.src8_frame_stride = frame_stride;
            continue;
        }
        q.I = (const float*)s.I[k].p; q.I_slot_stride = n;
        RcPyrArgs& p = pa[k];
        memset(&p, 0, sizeof(p));
        p.src = d_src; p.src_step = step; p.src_frame_stride = frame_stride;
        p.W0 = pl.w; p.H0 = pl.h;
        p.dst = (float*)s.I[k].p; p.dst_slot_stride = n;
        p.dslot0 = dslot0; p.nslots = pl.nslots; p.zstep = zstep;
        p.w = L.w; p.h = L.h; p.scale_x = L.scale_x; p.scale_y = L.scale_y;
        p.ksize = L.ksize; p.kern = (const float*)s.kern.p + pl.kern_off[k];
        p.tw = L.pyr_tw; p.th = L.pyr_th; p.reg_wp = L.pyr_reg_w; p.reg_hmax = L.pyr_reg_h;
        p.direct = (ctx->ablate & RC_ABL_PYR_STAGED) != 0;
        p.fixed3 = k == 0 && L.ksize == 3 && !(L.sigma > 0) && L.w == pl.w && L.h == pl.h;
    }
    auto npx = [&](int k) { return (double)pl.lv[k].w * pl.lv[k].h; };
    if (pl.exact) {
        // option "exact": every scale (0 included) through the bit-exact pyramid kernels, then the
        // expansion in upstream's operation order
        for (int k = 0; k < pl.nlev; k++) {
            { RcProfScope ps(ctx, s.cur, RC_K_PYR, k, (double)count * (npx(0) + 4. * npx(k)));
              rc_launch_pyr(pa[k], count, pl.lv[k].pyr_lds, s.cur); }
            RcProfScope ps(ctx, s.cur, RC_K_POLY, k, (double)count * 24. * npx(k));
            rc_launch_exact_polyexp(qa[k], count, s.cur);
        }
        RC_HIP(hipGetLastError());
        return RC_OK;
    }
    // A frame or two per call (the frame-at-a-time loop, two-image calls): the scales' grids are each
    // smaller than the GPU and independent of one another, so scales 1 + 2 of the pyramid share one launch
    // and the expansions of scales 0..2 another (block-index dispatch; same tile code, same bits).
    const bool merge = ctx->merge_small && count <= 2;
    // Larger batches at pyr_scale 0.5 with exact half / quarter sizes: the scale-0 expansion writes
    // pyramid scales 1 and 2 from the bytes it has staged anyway (option "fuse_pyr"; same bits)
    int npyr = 0;
    if (!merge && ctx->fuse_pyr && pl.nlev >= 2 && rc_polyexp_pyr_ok(qa[0])) {
        auto exact = [&](int k, int f) { return pl.lv[k].w * f == pl.w && pl.lv[k].h * f == pl.h; };
        if (exact(1, 2) && pl.lv[1].ksize == 3) {
            npyr = 1;
            if (pl.nlev >= 3 && exact(2, 4) && pl.lv[2].ksize == 9) npyr = 2;
        }
    }
    if (npyr) {
        RcPolyArgs& q = qa[0];
        q.npyr = npyr;
        double alg = 21. * npx(0), model = 29. * npx(0);
        for (int k = 1; k <= npyr; k++) {
            q.py[k - 1].dst = pa[k].dst; q.py[k - 1].dst_slot_stride = pa[k].dst_slot_stride;
            q.py[k - 1].w = pa[k].w; q.py[k - 1].h = pa[k].h; q.py[k - 1].kern = pa[k].kern;
            alg += 4. * npx(k); model += npx(0) + 4. * npx(k);
        }
        RcProfScope ps(ctx, s.cur, RC_K_POLY, 0, (double)count * alg, (double)count * model);
        rc_launch_polyexp(q, count, s.cur);
    }
    int k_pyr = 1 + npyr;
    if (merge && pl.nlev >= 3 && rc_pyr_pair_ok(pa[1], pa[2])) {
        RcProfScope ps(ctx, s.cur, RC_K_PYR, 1, (double)count * (2. * npx(0) + 4. * (npx(1) + npx(2))));
        rc_launch_pyr_pair(pa[1], pa[2], count, s.cur);
        k_pyr = 3;
    }
    for (int k = k_pyr; k < pl.nlev; k++) {
        RcProfScope ps(ctx, s.cur, RC_K_PYR, k, (double)count * (npx(0) + 4. * npx(k)));
        rc_launch_pyr(pa[k], count, pl.lv[k].pyr_lds, s.cur);
    }
    int k_poly = 0;
    const int nm = pl.nlev < 3 ? pl.nlev : 3;
    if (merge && nm >= 2 && rc_polyexp_multi_ok(qa, nm)) {
        double alg = 21. * npx(0), model = 29. * npx(0);
        for (int k = 1; k < nm; k++) { alg += 24. * npx(k); model += 24. * npx(k); }
        RcProfScope ps(ctx, s.cur, RC_K_POLY, 0, (double)count * alg, (double)count * model);
        rc_launch_polyexp_multi(qa, nm, count, s.cur);
        k_poly = nm;
    }
    if (npyr) k_poly = 1;
    for (int k = k_poly; k < pl.nlev; k++) {
        // SURVEY 8(d), scale 0: pyramid (N0 + 4 N0) + expansion (4 N0 + 20 N0) for the two stages fused there
        RcProfScope ps(ctx, s.cur, RC_K_POLY, k, (double)count * (k == 0 ? 21. : 24.) * npx(k),
                       k == 0 ? (double)count * 29. * npx(0) : -1.);
        rc_launch_polyexp(qa[k], count, s.cur);
    }
    RC_HIP(hipGetLastError());
    return RC_OK;
}

// Option "exact": the level driver over exact_kernels.hip (M and the window's column sums in HBM).
static int compute_flows_exact(rc_ctx* ctx, RcSlot& s, int pairs, int slot0, float* d_out, size_t out_pair_stride,
                               size_t out_step, int slot1, int zstep) {
    RcPlan& pl = s.plan;
    const int iters = pl.prm.iterations;
    const size_t n0 = (size_t)pl.lv[0].w * pl.lv[0].h;
    int rc;
    if ((rc = rc_buf_ensure(s.exM, n0 * 5 * pairs * sizeof(float)))) return rc;
    // (box windows: the column sums may be stored transposed with rows padded to a multiple of 16)
    const size_t n0p = (size_t)pl.lv[0].w * ((pl.lv[0].h + 15) & ~15);
    if ((rc = rc_buf_ensure(s.exV, pl.win.gaussian ? n0 * 5 * pairs * sizeof(float) : n0p * 5 * pairs * sizeof(double)))) return rc;
    if (!pl.win.gaussian && (rc = rc_buf_ensure(s.exG, n0 * 5 * pairs * sizeof(double)))) return rc;
    const float2* coarse = nullptr;
    int cw = 0, ch = 0;
    for (int k = pl.nlev - 1; k >= 0; k--) {
        const RcLevel& L = pl.lv[k];
        RcExactArgs a;
        memset(&a, 0, sizeof(a));
        a.RA = (const float4*)s.RA[k].p; a.RB = (const float*)s.RB[k].p; a.n = (size_t)L.w * L.h;
        a.slot0 = slot0; a.slot1 = slot1 >= 0 ? slot1 : (slot0 + 1) % pl.nslots; a.nslots = pl.nslots; a.zstep = zstep;
        a.w = L.w; a.h = L.h; a.win = pl.win; a.plain_scans = (ctx->ablate & RC_ABL_EXACT_PLAIN_SCANS) != 0;
        a.fused_matrices = (ctx->ablate & RC_ABL_EXACT_FUSED_M) != 0;
        a.flow = (float2*)s.FA[k].p; a.M = (float*)s.exM.p; a.V = s.exV.p; a.G = s.exG.p;
        if (coarse) {
            a.fin = coarse; a.fin_pair_stride = (size_t)cw * ch; a.fin_w = cw; a.fin_h = ch;
            a.up_scale_x = 1. / ((double)L.w / cw); a.up_scale_y = 1. / ((double)L.h / ch);
            a.up_mul = (float)(1. / pl.prm.pyr_scale);
        }
        const double nb = (double)pairs * a.n;
        { RcProfScope ps(ctx, s.cur, RC_K_ITER, k, 8. * nb + (coarse ? 8. * pairs * cw * ch : 0.));
          rc_launch_exact_flow_init(a, pairs, s.cur); }
        for (int i = 0; i < iters; i++) {
            if (k == 0 && i == iters - 1) { a.out = (char*)d_out; a.out_step = out_step; a.out_pair_stride = out_pair_stride; }
            if (rc_exact_iteration_fused_ok(a)) {
                // matrices inside the column scan: flow 8 + R0 20 + R1 20 -> V 40; V 40 -> flow 8
                RcProfScope ps(ctx, s.cur, RC_K_ITER, k, 136. * nb);
                rc_launch_exact_iteration_fused(a, pairs, s.cur);
                continue;
            }
            { RcProfScope ps(ctx, s.cur, RC_K_ITER, k, 68. * nb); rc_launch_exact_matrices(a, pairs, s.cur); }
            { RcProfScope ps(ctx, s.cur, RC_K_ITER, k, (pl.win.gaussian ? 68. : 108.) * nb);
              rc_launch_exact_window_solve(a, pairs, s.cur); }
        }
        if (k == 0 && iters == 0)
            for (int z = 0; z < pairs; z++)
                RC_HIP(hipMemcpy2DAsync((char*)d_out + (size_t)z * out_pair_stride, out_step, a.flow + (size_t)z * a.n,
                                        (size_t)L.w * 8, (size_t)L.w * 8, (size_t)L.h, hipMemcpyDeviceToDevice, s.cur));
        coarse = a.flow;
        cw = L.w; ch = L.h;
    }
    RC_HIP(hipGetLastError());
    return RC_OK;
}

// Coarse-to-fine flow for `pairs` frame pairs whose expansions sit in slots slot0+z, slot0+z+1.
static int compute_flows(rc_ctx* ctx, RcSlot& s, int pairs, int slot0, float* d_out, size_t out_pair_stride,
                         size_t out_step, int slot1 = -1, int zstep = 1) {
    if (!s.in_ts_push) s.ts_streak = 0;
    // option "exact": box windows replay upstream's running sums (exact_kernels.hip); Gaussian windows run the
    // kernels below from the build of flow_iter_kernels.hip that keeps upstream's operation order
    if (s.plan.exact && !s.plan.win.gaussian)
        return compute_flows_exact(ctx, s, pairs, slot0, d_out, out_pair_stride, out_step, slot1, zstep);
    const bool exact = s.plan.exact != 0;
    RcPlan& pl = s.plan;
    const int iters = pl.prm.iterations;
    const float2* coarse = nullptr;
    int cw = 0, ch = 0;
    for (int k = pl.nlev - 1; k >= 0; k--) {
        const RcLevel& L = pl.lv[k];
        size_t n = (size_t)L.w * L.h;
        RcIterArgs a;
        memset(&a, 0, sizeof(a));
        a.RA = (const float4*)s.RA[k].p; a.RB = (const float*)s.RB[k].p; a.R_slot_stride = n;
        a.slot0 = slot0; a.slot1 = slot1 >= 0 ? slot1 : (slot0 + 1) % pl.nslots; a.nslots = pl.nslots; a.zstep = zstep;
        a.w = L.w; a.h = L.h;
        a.win = pl.win;
        a.xcd_remap = ctx->xcd_remap;
        a.ablate = ctx->ablate;
        a.chain = ctx->chain;
        a.chain_min_blocks = ctx->chain_min_blocks;
        const float2* cur_in = nullptr;
        int passes = iters > 0 ? iters : 1;
        int nout = 0;   // intermediate buffers written so far at this scale (ping-pong index)
        for (int i = 0; i < passes;) {
            double in_bytes;
            if (i == 0) {
                if (!coarse) { a.in_mode = 0; a.fin = nullptr; in_bytes = 0; }
                else {
                    a.in_mode = 2; a.fin = coarse; a.fin_pair_stride = (size_t)cw * ch;
                    a.fin_w = cw; a.fin_h = ch;
                    a.up_scale_x = 1. / ((double)L.w / cw);
                    a.up_scale_y = 1. / ((double)L.h / ch);
                    a.up_mul = (float)(1. / pl.prm.pyr_scale);
                    a.up_exact2 = (L.w == 2 * cw && L.h == 2 * ch) ? 1 : 0;
                    in_bytes = 8. * cw * ch;
                }
            } else {
                a.in_mode = 1; a.fin = cur_in; a.fin_pair_stride = n; in_bytes = 8. * n;
            }
            a.solve = iters > 0 ? 1 : 0;
            // 32-bit offsets inside one frame's coefficient planes and one pair's flow field (the scale's own buffers and,
            // for the last launch of scale 0, the caller's rows): the condition of the fused two-iteration kernel
            a.addr32 = n * 16 < (1ull << 32) && (size_t)L.h * (k == 0 ? (out_step > (size_t)L.w * 8 ? out_step : (size_t)L.w * 8) : (size_t)L.w * 8) < (1ull << 32);
            // two iterations per launch whenever two are left and the window allows it
            int fuse = (passes - i >= 2 && ctx->fuse_iters && rc_flow_fast::rc_flow_iter_can_fuse2(a)) ? 2 : 1;
            bool last = (i + fuse == passes);
            if (last && k == 0) {
                a.fout = (char*)d_out; a.fout_step = out_step; a.fout_pair_stride = out_pair_stride;
            } else {
                float2* dst = (float2*)((nout & 1) ? s.FB[k].p : s.FA[k].p);
                nout++;
                a.fout = (char*)dst; a.fout_step = (size_t)L.w * 8; a.fout_pair_stride = n * 8;
                cur_in = dst;
            }
            {
                // SURVEY 8(d) bytes of the stages this launch stands for: the first launch of a scale
                // carries "init matrices" (8 N_{k+1} + 60 N_k), every iteration but the last 80 N_k
                // (blur+solve fused with the next matrix update), the last one 28 N_k
                double model = 0;
                if (i == 0) model += 60. * n + (coarse ? 8. * cw * ch : 0.);
                for (int j = i; j < i + fuse; j++) model += (j == passes - 1) ? 28. * n : 80. * n;
                // compulsory bytes as built: R1 20 B/px per pair, R0 20 B/px per pair that reads it from memory (the head of
                // a tile chain; the rest take it from the previous pair's LDS window), flow in, flow out
                const int r0 = fuse != 2 ? pairs : exact ? rc_flow_exact::rc_flow_iter2_r0_reads(a, pairs) : rc_flow_fast::rc_flow_iter2_r0_reads(a, pairs);
                RcProfScope ps(ctx, s.cur, fuse == 2 ? RC_K_ITER2 : RC_K_ITER, k,
                               (a.solve ? 20. * n * (pairs + r0) : 0.) + (double)pairs * (in_bytes + 8. * n), (double)pairs * model);
                if (exact) {
                    if (fuse == 2) rc_flow_exact::rc_launch_flow_iter2(a, pairs, s.cur);
                    else rc_flow_exact::rc_launch_flow_iter(a, pairs, s.cur);
                } else {
                    if (fuse == 2) rc_flow_fast::rc_launch_flow_iter2(a, pairs, s.cur);
                    else rc_flow_fast::rc_launch_flow_iter(a, pairs, s.cur);
                }
            }
            i += fuse;
        }
        coarse = cur_in;
        cw = L.w; ch = L.h;
    }
    RC_HIP(hipGetLastError());
    return RC_OK;
}

// ---------------------------------------------------------------------------- A entry points
extern "C" int rcflow_farneback_dev(rc_ctx* ctx, int stream, const uint8_t* d_prev, size_t prev_step,
                                    const uint8_t* d_next, size_t next_step, int w, int h, float* d_flow,
                                    size_t flow_step, const rc_farneback_params* p) {
    RcSlot* s = rc_slot(ctx, stream);
    if (!s || !d_prev || !d_next || !d_flow) { if (s) rc_set_error("null image pointer"); return RC_EINVAL; }
    if (prev_step < (size_t)w || next_step < (size_t)w || flow_step < (size_t)w * 8) {
        rc_set_error("row step smaller than a row");
        return RC_EINVAL;
    }
    RC_HIP(hipSetDevice(ctx->device));
    int rc = ensure_plan(ctx, *s, w, h, p, ctx->chunk);
    if (rc) return rc;
    s->primed = 0;
    s->batch_primed = 0;
    s->flow_w = s->flow_h = 0;      // the stream starts over: no resident flow field of its own (rcflow_stream_flow_ptr)
    if ((rc = expand_frames(ctx, *s, d_prev, 0, prev_step, 1, 0))) return rc;
    if ((rc = expand_frames(ctx, *s, d_next, 0, next_step, 1, 1))) return rc;
    return compute_flows(ctx, *s, 1, 0, d_flow, 0, flow_step);
}

extern "C" int rcflow_farneback_u8(rc_ctx* ctx, int stream, const uint8_t* prev, size_t prev_step,
                                   const uint8_t* next, size_t next_step, int w, int h, float* flow,
                                   size_t flow_step, double pyr_scale, int levels, int winsize, int iterations,
                                   int poly_n, double poly_sigma, int flags) {
    RcSlot* s = rc_slot(ctx, stream);
    if (!s || !prev || !next || !flow || w <= 0 || h <= 0) { if (s) rc_set_error("null image pointer"); return RC_EINVAL; }
    if (prev_step < (size_t)w || next_step < (size_t)w || flow_step < (size_t)w * 8) {
        rc_set_error("row step smaller than a row");
        return RC_EINVAL;
    }
    rc_farneback_params p = {pyr_scale, levels, winsize, iterations, poly_n, poly_sigma, flags};
    RC_HIP(hipSetDevice(ctx->device));
    int rc = ensure_plan(ctx, *s, w, h, &p, ctx->chunk);
    if (rc) return rc;
    size_t fb = (size_t)w * h;
    if ((rc = rc_buf_ensure(s->stage_u8, 2 * fb))) return rc;
    if ((rc = rc_buf_ensure(s->stage_flow, fb * 8))) return rc;
    uint8_t* du = (uint8_t*)s->stage_u8.p;
    RC_HIP(hipMemcpy2DAsync(du, w, prev, prev_step, w, h, hipMemcpyHostToDevice, s->cur));
    RC_HIP(hipMemcpy2DAsync(du + fb, w, next, next_step, w, h, hipMemcpyHostToDevice, s->cur));
    rc = rcflow_farneback_dev(ctx, stream, du, w, du + fb, w, w, h, (float*)s->stage_flow.p, (size_t)w * 8, &p);
    if (rc) return rc;
    RC_HIP(hipMemcpy2DAsync(flow, flow_step, s->stage_flow.p, (size_t)w * 8, (size_t)w * 8, h,
                            hipMemcpyDeviceToHost, s->cur));
    RC_HIP(hipStreamSynchronize(s->cur));
    return RC_OK;
}

extern "C" int rcflow_stream_reset(rc_ctx* ctx, int stream) {
    RcSlot* s = rc_slot(ctx, stream);
    if (!s) return RC_EINVAL;
    s->primed = 0;
    s->cur_slot = 0;
    s->flow_w = s->flow_h = 0;
    return RC_OK;
}

// Frame loop on two streams.  The expansion of a frame depends only on the frame, the flow on the expansions of
// this frame and the previous one: with `pre` set the caller-supplied step (an upload) and the expansion run on the
// slot's second stream, beside the flow kernels of the PREVIOUS frame that the slot's stream may still be executing
// (the calls are asynchronous, so a host that pushes ahead of the GPU gets the overlap; at one frame per call the
// coarse scales' grids are smaller than the GPU and the two really run side by side).  Order kept by events:
//   second stream: waits for the flow launches of the push before the previous one (every ring slot and frame buffer of
//                  that age is free), [upload], expansion -> `expanded`
//   slot's stream: waits for `expanded`, flow launches -> flow_done
struct RcFrameAux {
    const void* host_src; void* d_dst; size_t bytes;     // optional upload executed on the second stream
    hipEvent_t after_upload;                             // recorded behind the upload (staging buffer free again)
};

static int push_frame_core(rc_ctx* ctx, RcSlot* s, int stream, const uint8_t* d_frame, size_t step, int w, int h,
                           float* d_flow, size_t flow_step, const rc_farneback_params* p, bool two_streams,
                           const RcFrameAux* up) {
    int was_valid = s->plan.valid;
    int rc = ensure_plan(ctx, *s, w, h, p, ctx->chunk);
    if (rc) return rc;
    if (!was_valid) s->primed = 0;
    s->batch_primed = 0;
    (void)stream;
    auto upload_on = [&](hipStream_t st) -> int {
        if (!up) return RC_OK;
        RC_HIP(hipMemcpyAsync(up->d_dst, up->host_src, up->bytes, hipMemcpyHostToDevice, st));
        RC_HIP(hipEventRecord(up->after_upload, st));
        return RC_OK;
    };
    if (!s->primed) {
        if ((rc = upload_on(s->cur))) return rc;
        if ((rc = expand_frames(ctx, *s, d_frame, 0, step, 1, 0))) return rc;
        s->primed = 1;
        s->cur_slot = 0;
        return 1;
    }
    if (!d_flow || flow_step < (size_t)w * 8) { rc_set_error("bad flow buffer"); return RC_EINVAL; }
    const int nxt = (s->cur_slot + 1) % s->plan.nslots;
    if (two_streams && !ctx->prof_on && s->plan.nslots >= 4) {
        if (!s->aux) RC_HIP(hipStreamCreateWithFlags(&s->aux, hipStreamNonBlocking));
        for (auto& e : s->flow_done) if (!e) { RC_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming)); RC_HIP(hipEventRecord(e, s->cur)); }
        hipStream_t main_stream = s->cur;
        const int older = s->flow_done_i;                  // written two pushes ago
        if (s->ts_streak == 0) {
            // The previous operation on the slot was not a two-stream push (priming, a clip or pair call, the
            // one-stream path, a reset): its kernels may still read the ring slot, the scale images and the frame
            // buffers this push is about to overwrite, and no flow_done event covers them.  Join the slot's stream
            // once; from the next push on the flow_done events carry the order.
            hipEvent_t& joined = s->fev[s->fev_i];
            s->fev_i = (s->fev_i + 1) % 8;
            if (!joined) RC_HIP(hipEventCreateWithFlags(&joined, hipEventDisableTiming));
            RC_HIP(hipEventRecord(joined, main_stream));
            RC_HIP(hipStreamWaitEvent(s->aux, joined, 0));
        } else {
            RC_HIP(hipStreamWaitEvent(s->aux, s->flow_done[older], 0));
        }
        hipEvent_t& expanded = s->fev[s->fev_i];
        s->fev_i = (s->fev_i + 1) % 8;
        if (!expanded) RC_HIP(hipEventCreateWithFlags(&expanded, hipEventDisableTiming));
        if ((rc = upload_on(s->aux))) return rc;
        s->in_ts_push = true;
        s->cur = s->aux;
        rc = expand_frames(ctx, *s, d_frame, 0, step, 1, nxt);
        s->cur = main_stream;
        if (rc) { s->in_ts_push = false; return rc; }
        RC_HIP(hipEventRecord(expanded, s->aux));
        RC_HIP(hipStreamWaitEvent(main_stream, expanded, 0));
        rc = compute_flows(ctx, *s, 1, s->cur_slot, d_flow, 0, flow_step);
        s->in_ts_push = false;
        if (rc) return rc;
        RC_HIP(hipEventRecord(s->flow_done[older], main_stream));
        s->flow_done_i = older ^ 1;
        s->ts_streak++;
        s->cur_slot = nxt;
        return RC_OK;
    }
    if ((rc = upload_on(s->cur))) return rc;
    if ((rc = expand_frames(ctx, *s, d_frame, 0, step, 1, nxt))) return rc;
    if ((rc = compute_flows(ctx, *s, 1, s->cur_slot, d_flow, 0, flow_step))) return rc;
    s->cur_slot = nxt;
    return RC_OK;
}

extern "C" int rcflow_push_frame_dev(rc_ctx* ctx, int stream, const uint8_t* d_frame, size_t step, int w, int h,
                                     float* d_flow, size_t flow_step, const rc_farneback_params* p) {
    RcSlot* s = rc_slot(ctx, stream);
    if (!s || !d_frame) { if (s) rc_set_error("null frame pointer"); return RC_EINVAL; }
    if (step < (size_t)w) { rc_set_error("row step smaller than a row"); return RC_EINVAL; }
    RC_HIP(hipSetDevice(ctx->device));
    // option "frame_overlap" = 2: the caller guarantees that d_frame is complete when the call is made (a resident clip, a
    // producer it has synchronised) -- only then may the expansion start without waiting for the slot's stream
    return push_frame_core(ctx, s, stream, d_frame, step, w, h, d_flow, flow_step, p, ctx->frame_overlap >= 2, nullptr);
}

// The reference's frame loop with HOST frames (ripcurrents.cpp:198-221: video.read -> resize -> cvtColor ->
// copyTo(UMat) -> calcOpticalFlowFarneback -> copyTo(u_f2)): the frame goes through one of two page-locked
// staging buffers and is uploaded asynchronously; the flow field stays on the device for the analysis calls
// (rcflow_stream_flow_ptr) and only crosses PCIe when the host asks for it (rcflow_stream_flow_read).  The call
// returns once the frame is in the staging buffer -- the upload of frame t and the kernels of frame t run
// while the host decodes frame t + 1.  Returns 1 when the call only primed the stream (no flow yet).
// The slot's two page-locked staging buffers for frames of w x h bytes; returns the one the next push uses, once the
// upload that last read it has left it.
static int frame_staging_next(rc_ctx* ctx, RcSlot* s, int w, int h, int* idx) {
    if (w <= 0 || h <= 0) { rc_set_error("bad frame arguments"); return RC_EINVAL; }
    if (w > ctx->max_w || h > ctx->max_h) { rc_set_error("frame %dx%d exceeds the context's %dx%d", w, h, ctx->max_w, ctx->max_h); return RC_ESIZE; }
    RC_HIP(hipSetDevice(ctx->device));
    const size_t fb = (size_t)w * h;
    if (s->pin_bytes < fb) {
        RC_HIP(hipStreamSynchronize(s->cur));
        for (int i = 0; i < 2; i++) {
            if (s->pin[i]) (void)hipHostFree(s->pin[i]);
            s->pin[i] = nullptr;
            if (hipHostMalloc(&s->pin[i], fb, hipHostMallocDefault) != hipSuccess) { rc_set_error("hipHostMalloc(%zu) failed", fb); s->pin_bytes = 0; return RC_ENOMEM; }
            if (!s->pin_free[i]) RC_HIP(hipEventCreateWithFlags(&s->pin_free[i], hipEventDisableTiming));
            RC_HIP(hipEventRecord(s->pin_free[i], s->cur));
        }
        s->pin_bytes = fb;
    }
    int rc;
    if ((rc = rc_buf_ensure(s->stage_u8, 2 * fb))) return rc;
    if ((rc = rc_buf_ensure(s->stage_flow, fb * 8))) return rc;
    RC_HIP(hipEventSynchronize(s->pin_free[s->pin_i]));       // the upload that last used this staging buffer has left it
    *idx = s->pin_i;
    return RC_OK;
}

// Upload of staging buffer i (dense w x h bytes) + one step of the frame loop
static int push_staged_frame(rc_ctx* ctx, RcSlot* s, int stream, int i, int w, int h, const rc_farneback_params* p) {
    const size_t fb = (size_t)w * h;
    uint8_t* d_frame = (uint8_t*)s->stage_u8.p + (size_t)i * fb;
    s->pin_i = i ^ 1;
    s->pin_acq = -1;
    // the upload belongs to the expansion's side of the two-stream frame loop (option "frame_overlap" >= 1)
    RcFrameAux up = {s->pin[i], d_frame, fb, s->pin_free[i]};
    const int rc = push_frame_core(ctx, s, stream, d_frame, w, w, h, (float*)s->stage_flow.p, (size_t)w * 8, p, ctx->frame_overlap >= 1, &up);
    if (rc == RC_OK) { s->flow_w = w; s->flow_h = h; }
    else if (rc == 1) { s->flow_w = s->flow_h = 0; }
    return rc;
}

extern "C" int rcflow_push_frame_u8(rc_ctx* ctx, int stream, const uint8_t* frame, size_t step, int w, int h,
                                    const rc_farneback_params* p) {
    RcSlot* s = rc_slot(ctx, stream);
    if (!s || !frame || w <= 0 || h <= 0 || step < (size_t)w) { if (s) rc_set_error("bad frame arguments"); return RC_EINVAL; }
    int i, rc;
    if ((rc = frame_staging_next(ctx, s, w, h, &i))) return rc;
    uint8_t* dst = (uint8_t*)s->pin[i];
    if (step == (size_t)w) memcpy(dst, frame, (size_t)w * h);
    else for (int y = 0; y < h; y++) memcpy(dst + (size_t)y * w, frame + (size_t)y * step, w);
    return push_staged_frame(ctx, s, stream, i, w, h, p);
}

// The same loop without the copy: the host produces the frame INTO the staging buffer (e.g. as the destination of the
// cvtColor at ripcurrents.cpp:210) and then pushes it.
extern "C" int rcflow_frame_buffer_acquire(rc_ctx* ctx, int stream, int w, int h, uint8_t** host_frame, size_t* step) {
    RcSlot* s = rc_slot(ctx, stream);
    if (!s || !host_frame) { if (s) rc_set_error("bad frame arguments"); return RC_EINVAL; }
    int i, rc;
    if ((rc = frame_staging_next(ctx, s, w, h, &i))) return rc;
    s->pin_acq = i; s->pin_w = w; s->pin_h = h;
    *host_frame = (uint8_t*)s->pin[i];
    if (step) *step = (size_t)w;
    return RC_OK;
}
extern "C" int rcflow_push_frame_acquired(rc_ctx* ctx, int stream, const rc_farneback_params* p) {
    RcSlot* s = rc_slot(ctx, stream);
    if (!s) return RC_EINVAL;
    if (s->pin_acq < 0 || s->pin_acq != s->pin_i) { rc_set_error("no frame buffer acquired (rcflow_frame_buffer_acquire) since the last push"); return RC_ESTATE; }
    RC_HIP(hipSetDevice(ctx->device));
    return push_staged_frame(ctx, s, stream, s->pin_acq, s->pin_w, s->pin_h, p);
}

// Device address of the flow field the last rcflow_push_frame_u8 produced (w x h float2, dense rows), for the
// analysis entry points; valid until the next push on the slot.
extern "C" int rcflow_stream_flow_ptr(rc_ctx* ctx, int stream, float** d_flow, int* w, int* h) {
    RcSlot* s = rc_slot(ctx, stream);
    if (!s || !d_flow) return RC_EINVAL;
    if (!s->flow_w) { rc_set_error("no flow field yet: the stream has only been primed"); return RC_ESTATE; }
    *d_flow = (float*)s->stage_flow.p;
    if (w) *w = s->flow_w;
    if (h) *h = s->flow_h;
    return RC_OK;
}

// Copies that flow field to the host (CV_32FC2 layout, byte step) and waits for it.
extern "C" int rcflow_stream_flow_read(rc_ctx* ctx, int stream, float* flow, size_t flow_step) {
    RcSlot* s = rc_slot(ctx, stream);
    if (!s || !flow) return RC_EINVAL;
    if (!s->flow_w) { rc_set_error("no flow field yet: the stream has only been primed"); return RC_ESTATE; }
    if (flow_step < (size_t)s->flow_w * 8) { rc_set_error("row step smaller than a row"); return RC_EINVAL; }
    RC_HIP(hipSetDevice(ctx->device));
    RC_HIP(hipMemcpy2DAsync(flow, flow_step, s->stage_flow.p, (size_t)s->flow_w * 8, (size_t)s->flow_w * 8, s->flow_h,
                            hipMemcpyDeviceToHost, s->cur));
    RC_HIP(hipStreamSynchronize(s->cur));
    return RC_OK;
}

// Streaming form of the clip call: the frames CONTINUE the slot's stream (the one rcflow_push_frame_dev
// feeds), so a segment processed in batches expands every frame exactly once.  Returns the number of flow
// fields written to d_flows[0 ..): nframes when the stream was primed (flow 0 = previous call's last frame ->
// d_frames[0]), nframes - 1 when this call primed it (first call after rcflow_stream_reset, or another size /
// other parameters); negative RC_E* on error.
extern "C" int rcflow_push_clip_dev(rc_ctx* ctx, int stream, const uint8_t* d_frames, size_t frame_stride, size_t step,
                                    int nframes, int w, int h, float* d_flows, size_t flow_frame_stride,
                                    size_t flow_step, const rc_farneback_params* p) {
    RcSlot* s = rc_slot(ctx, stream);
    if (!s || !d_frames || nframes < 1) { if (s) rc_set_error("bad clip arguments"); return RC_EINVAL; }
    if (step < (size_t)w || (nframes > 1 && frame_stride < step * (size_t)(h - 1) + w)) {
        rc_set_error("clip strides smaller than a frame");
        return RC_EINVAL;
    }
    RC_HIP(hipSetDevice(ctx->device));
    int rc = ensure_plan(ctx, *s, w, h, p, ctx->chunk);          // a new size / parameter set drops `primed`
    if (rc) return rc;
    s->batch_primed = 0;
    const int C = s->plan.chunk, ns = s->plan.nslots;
    int t = 0, s0 = s->cur_slot;
    if (!s->primed) {
        if ((rc = expand_frames(ctx, *s, d_frames, 0, step, 1, 0))) return rc;
        s->primed = 1;
        s->cur_slot = s0 = 0;
        t = 1;
    }
    const int nflows = nframes - t;
    if (nflows > 0 && (!d_flows || flow_step < (size_t)w * 8 ||
                       (nflows > 1 && flow_frame_stride < flow_step * (size_t)(h - 1) + (size_t)w * 8))) {
        rc_set_error("bad flow buffer");
        return RC_EINVAL;
    }
    for (int done = 0; t < nframes;) {
        const int np = nframes - t < C ? nframes - t : C;
        if ((rc = expand_frames(ctx, *s, d_frames + (size_t)t * frame_stride, frame_stride, step, np, (s0 + 1) % ns))) {
            s->primed = 0;
            return rc;
        }
        if ((rc = compute_flows(ctx, *s, np, s0, (float*)((char*)d_flows + (size_t)done * flow_frame_stride),
                                flow_frame_stride, flow_step))) {
            s->primed = 0;
            return rc;
        }
        s0 = (s0 + np) % ns;
        s->cur_slot = s0;
        t += np;
        done += np;
    }
    return nflows;
}

extern "C" int rcflow_farneback_clip_dev(rc_ctx* ctx, int stream, const uint8_t* d_frames, size_t frame_stride,
                                         size_t step, int nframes, int w, int h, float* d_flows,
                                         size_t flow_frame_stride, size_t flow_step,
                                         const rc_farneback_params* p) {
    RcSlot* s = rc_slot(ctx, stream);
    if (!s || !d_frames || !d_flows || nframes < 2) { if (s) rc_set_error("bad clip arguments"); return RC_EINVAL; }
    if (step < (size_t)w || flow_step < (size_t)w * 8 || frame_stride < step * (size_t)(h - 1) + w ||
        flow_frame_stride < flow_step * (size_t)(h - 1) + (size_t)w * 8) {
        rc_set_error("clip strides smaller than a frame");
        return RC_EINVAL;
    }
    RC_HIP(hipSetDevice(ctx->device));
    // Two chunks of expansions in the ring when the clip spans several chunks: the expansions of
    // chunk c+1 (VALU-bound) then run on a second stream beside the flow kernels of chunk c
    // (bound by the memory system), joined by events.  Results do not depend on it.
    const bool overlap = ctx->overlap && nframes - 1 > ctx->chunk;
    int rc = ensure_plan(ctx, *s, w, h, p, ctx->chunk, overlap ? 2 * ctx->chunk + 1 : 0);
    if (rc) return rc;
    s->primed = 0;
    s->batch_primed = 0;
    const int C = s->plan.chunk, ns = s->plan.nslots;
    int s0 = 0;
    if (!overlap) {
        for (int t = 0; t < nframes - 1;) {
            int np = nframes - 1 - t < C ? nframes - 1 - t : C;
            // the first chunk expands its np + 1 frames in one launch (the ring holds chunk + 1 slots);
            // later chunks reuse the last expansion of the previous one
            if (t == 0) rc = expand_frames(ctx, *s, d_frames, frame_stride, step, np + 1, 0);
            else rc = expand_frames(ctx, *s, d_frames + (size_t)(t + 1) * frame_stride, frame_stride, step, np, (s0 + 1) % ns);
            if (rc)
                return rc;
            if ((rc = compute_flows(ctx, *s, np, s0, (float*)((char*)d_flows + (size_t)t * flow_frame_stride),
                                    flow_frame_stride, flow_step)))
                return rc;
            s0 = (s0 + np) % ns;
            t += np;
        }
        return RC_OK;
    }
    if (!s->aux) RC_HIP(hipStreamCreateWithFlags(&s->aux, hipStreamNonBlocking));
    hipStream_t flow_stream = s->cur, exp_stream = s->aux;
    std::vector<hipEvent_t> evs;
    auto new_event = [&](hipStream_t st) -> hipEvent_t {
        hipEvent_t e = nullptr;
        if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return nullptr;
        (void)hipEventRecord(e, st);
        evs.push_back(e);
        return e;
    };
    auto cleanup = [&]() { for (hipEvent_t e : evs) (void)hipEventDestroy(e); };   // destruction is deferred by the runtime
    hipEvent_t fork = new_event(flow_stream);               // everything the caller queued so far (the frames)
    if (!fork) { rc_set_error("hipEventCreate failed"); return RC_EHIP; }
    (void)hipStreamWaitEvent(exp_stream, fork, 0);
    hipEvent_t flows_done[2] = {nullptr, nullptr};          // flows of chunk c-1, c-2 (their slots get reused)
    s->cur = exp_stream;
    rc = expand_frames(ctx, *s, d_frames, frame_stride, step, 1, 0);
    s->cur = flow_stream;
    for (int t = 0, c = 0; rc == RC_OK && t < nframes - 1; c++) {
        int np = nframes - 1 - t < C ? nframes - 1 - t : C;
        if (flows_done[c & 1]) (void)hipStreamWaitEvent(exp_stream, flows_done[c & 1], 0);     // chunk c-2 released its slots
        s->cur = exp_stream;
        rc = expand_frames(ctx, *s, d_frames + (size_t)(t + 1) * frame_stride, frame_stride, step, np, (s0 + 1) % ns);
        s->cur = flow_stream;
        if (rc) break;
        hipEvent_t expanded = new_event(exp_stream);
        if (!expanded) { rc = RC_EHIP; break; }
        (void)hipStreamWaitEvent(flow_stream, expanded, 0);
        rc = compute_flows(ctx, *s, np, s0, (float*)((char*)d_flows + (size_t)t * flow_frame_stride), flow_frame_stride,
                           flow_step);
        if (rc) break;
        if (!(flows_done[c & 1] = new_event(flow_stream))) { rc = RC_EHIP; break; }
        s0 = (s0 + np) % ns;
        t += np;
    }
    cleanup();
    return rc;
}

// ---------------------------------------------------------------------------- the whole frame loop, one launch per frame
void rc_loop_graph_drop(RcSlot& s) {
    for (int i = 0; i < 2; i++) {
        if (s.loop_exec[i]) (void)hipGraphExecDestroy((hipGraphExec_t)s.loop_exec[i]);
        s.loop_exec[i] = nullptr;
        s.loop_eager[i] = 0;
    }
}

struct RcLoopKey {          // everything a captured launch sequence has baked in
    rc_frame_loop loop;
    int w, h, pin, ring, ablate, chain;
    void* hip_stream;
    void* d_frame; void* pin_host; void* d_flow;
};
static_assert(sizeof(RcLoopKey) <= sizeof(((RcSlot*)nullptr)->loop_key[0]), "RcSlot::loop_key too small");

// the analysis chain of one frame on the resident flow field (ripcurrents.cpp:229-479), on s.cur
static int loop_analysis(rc_ctx* ctx, RcSlot& s, int stream, const rc_frame_loop& L, int w, int h) {
    const size_t fs = (size_t)w * 8;
    float* d_flow = (float*)s.stage_flow.p;
    int rc;
    if ((rc = rc_loop_counter(ctx, s, false, 0))) return rc;                                          // framecount++ (ripcurrents.cpp:194)
    if ((rc = rcflow_advect_field_dev(ctx, stream, d_flow, fs, w, h, L.dt, L.iterations, -1.f))) return rc;     // :229-231, last frame's UPPER
    if (L.nseeds > 0 && (rc = rcflow_advect_points_dev(ctx, stream, L.d_seeds, L.nseeds, d_flow, fs, w, h, L.seed_dt, L.seed_iterations,
                                                       L.seed_upper, L.seed_variant, nullptr))) return rc;      // :283-285
    if ((rc = rcflow_histogram_dev(ctx, stream, d_flow, fs, w, h))) return rc;                          // :319-330
    if ((rc = rcflow_thresholds_dev(ctx, stream))) return rc;                                           // :333-366
    if ((rc = rc_classify_accumulate(ctx, stream, d_flow, fs, w, h, -1, L.MID, L.LOWER, nullptr, 0, nullptr, 0, nullptr, 0,
                                     L.d_outmask, L.mask_step))) return rc;                               // :376-439
    if (L.d_edges && (rc = rcflow_create_edges_dev(ctx, stream, L.d_outmask, L.mask_step, w, h, L.d_edges, L.edges_step))) return rc;   // :477-479
    return RC_OK;
}

// upload + expansion + flow + the analysis chain of one frame on s.cur (eager or under stream capture)
static int loop_launches(rc_ctx* ctx, RcSlot& s, int stream, const rc_frame_loop& L, int pin, int w, int h) {
    const size_t fb = (size_t)w * h, fs = (size_t)w * 8;
    uint8_t* d_frame = (uint8_t*)s.stage_u8.p + (size_t)pin * fb;
    float* d_flow = (float*)s.stage_flow.p;
    const int cur = s.cur_slot, nxt = cur ^ 1;
    RC_HIP(hipMemcpyAsync(d_frame, s.pin[pin], fb, hipMemcpyHostToDevice, s.cur));
    int rc;
    if ((rc = expand_frames(ctx, s, d_frame, 0, w, 1, nxt))) return rc;
    if ((rc = compute_flows(ctx, s, 1, cur, d_flow, 0, fs, nxt))) return rc;
    return loop_analysis(ctx, s, stream, L, w, h);
}

static int push_staged_frame(rc_ctx* ctx, RcSlot* s, int stream, int i, int w, int h, const rc_farneback_params* p);

extern "C" int rcflow_frame_loop_step(rc_ctx* ctx, int stream, const rc_farneback_params* p, const rc_frame_loop* loop) {
    RcSlot* s = rc_slot(ctx, stream);
    if (!s || !p || !loop) { if (s) rc_set_error("null argument"); return RC_EINVAL; }
    if (s->pin_acq < 0 || s->pin_acq != s->pin_i) { rc_set_error("no frame buffer acquired (rcflow_frame_buffer_acquire) since the last push"); return RC_ESTATE; }
    if (loop->nseeds < 0 || (loop->nseeds && !loop->d_seeds) || (loop->d_edges && !loop->d_outmask) || loop->iterations < 0) {
        rc_set_error("bad frame-loop configuration");
        return RC_EINVAL;
    }
    RC_HIP(hipSetDevice(ctx->device));
    const int w = s->pin_w, h = s->pin_h, pin = s->pin_acq;
    int rc;
    // the histogram stage's int32 guard, checked before anything is consumed or launched: a refused call leaves the
    // acquired frame, the stream and the analysis state as they were (reset the segment, call again)
    if (s->primed && s->an.w == w && s->an.h == h && (rc = rc_hist_book(*s, w, h, false))) return rc;
    if (!loop->use_graph) {
        // The default: the two-stream frame push (upload and expansion of this frame beside the previous frame's flow
        // and analysis kernels, which the slot's stream may still be executing) followed by the analysis launches --
        // measured faster than one linear captured sequence per frame, which cannot overlap across frames
        if ((rc = rc_analysis_ensure(ctx, *s, w, h))) return rc;
        rc = push_staged_frame(ctx, s, stream, pin, w, h, p);
        if (rc < 0) return rc;
        if (rc == 1) {
            s->loop_fc = 0;
            if ((rc = rc_loop_counter(ctx, *s, true, 0))) return rc;
            return 1;
        }
        if ((rc = loop_analysis(ctx, *s, stream, *loop, w, h))) return rc;
        s->loop_fc++;
        return RC_OK;
    }
    const bool was_valid = s->plan.valid;
    rc = ensure_plan(ctx, *s, w, h, p, 1, 2);                // a ring of two expansions: the captured sequences alternate
    if (rc) return rc;
    if (!was_valid) s->primed = 0;
    s->batch_primed = 0;
    if ((rc = rc_analysis_ensure(ctx, *s, w, h))) return rc;
    const size_t fb = (size_t)w * h;
    if (!s->primed) {
        rc_loop_graph_drop(*s);
        uint8_t* d_frame = (uint8_t*)s->stage_u8.p + (size_t)pin * fb;
        RC_HIP(hipMemcpyAsync(d_frame, s->pin[pin], fb, hipMemcpyHostToDevice, s->cur));
        RC_HIP(hipEventRecord(s->pin_free[pin], s->cur));
        if ((rc = expand_frames(ctx, *s, d_frame, 0, w, 1, 0))) return rc;
        s->loop_fc = 0;
        if ((rc = rc_loop_counter(ctx, *s, true, 0))) return rc;
        s->primed = 1; s->cur_slot = 0;
        s->pin_i = pin ^ 1; s->pin_acq = -1;
        s->flow_w = s->flow_h = 0;
        return 1;
    }
    const int cur = s->cur_slot;
    if (cur > 1) { rc_set_error("the slot's ring is not the frame loop's: call rcflow_stream_reset first"); return RC_ESTATE; }
    RcLoopKey key;
    memset(&key, 0, sizeof(key));
    key.loop = *loop;
    key.w = w; key.h = h; key.pin = pin; key.ring = cur; key.ablate = ctx->ablate; key.chain = ctx->chain;
    key.hip_stream = (void*)s->cur;
    key.d_frame = (uint8_t*)s->stage_u8.p + (size_t)pin * fb; key.pin_host = s->pin[pin]; key.d_flow = s->stage_flow.p;
    const bool graph_ok = !ctx->prof_on && s->cur != nullptr;
    const bool same = !memcmp(&key, s->loop_key[cur], sizeof(key));
    if (graph_ok && s->loop_exec[cur] && same) {
        // steady state: one launch replays the frame's whole sequence; what the eager calls book on the host is booked here
        if ((rc = rc_hist_book(*s, w, h, true))) return rc;
        RC_HIP(hipGraphLaunch((hipGraphExec_t)s->loop_exec[cur], s->cur));
        s->ts_streak = 0;
    } else if (graph_ok && s->loop_eager[cur] && same) {
        // second frame of this parity with the same configuration: capture the sequence while issuing it, then launch it
        if ((rc = rc_hist_book(*s, w, h, false))) return rc;
        hipGraph_t graph = nullptr;
        RC_HIP(hipStreamBeginCapture(s->cur, hipStreamCaptureModeRelaxed));
        rc = loop_launches(ctx, *s, stream, *loop, pin, w, h);
        hipError_t e = hipStreamEndCapture(s->cur, &graph);
        if (rc) { if (graph) (void)hipGraphDestroy(graph); return rc; }
        if (e != hipSuccess || !graph) { rc_set_error("hipStreamEndCapture failed: %s", hipGetErrorString(e)); return RC_EHIP; }
        hipGraphExec_t exec = nullptr;
        e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
        (void)hipGraphDestroy(graph);
        if (e != hipSuccess) { rc_set_error("hipGraphInstantiate failed: %s", hipGetErrorString(e)); return RC_EHIP; }
        s->loop_exec[cur] = exec;
        RC_HIP(hipGraphLaunch(exec, s->cur));
    } else {
        if (s->loop_exec[cur]) { (void)hipGraphExecDestroy((hipGraphExec_t)s->loop_exec[cur]); s->loop_exec[cur] = nullptr; }
        if ((rc = loop_launches(ctx, *s, stream, *loop, pin, w, h))) return rc;
        memcpy(s->loop_key[cur], &key, sizeof(key));
        s->loop_eager[cur] = 1;
    }
    RC_HIP(hipEventRecord(s->pin_free[pin], s->cur));        // (behind the whole frame: the staging buffer comes round two frames later)
    s->cur_slot = cur ^ 1;
    s->pin_i = pin ^ 1; s->pin_acq = -1;
    s->flow_w = w; s->flow_h = h;
    s->loop_fc++;
    return RC_OK;
}

// ---------------------------------------------------------------------------- lockstep batch of streams
void rc_batch_graph_drop(RcSlot& s) {
    for (int i = 0; i < 2; i++) {
        if (s.batch_exec[i]) (void)hipGraphExecDestroy((hipGraphExec_t)s.batch_exec[i]);
        s.batch_exec[i] = nullptr;
        s.batch_eager[i] = 0;
    }
}

static int batch_launches(rc_ctx* ctx, RcSlot& s, const uint8_t* d_frames, size_t frame_stride, size_t step, int S,
                          float* d_flows, size_t flow_frame_stride, size_t flow_step, int cur) {
    int rc = expand_frames(ctx, s, d_frames, frame_stride, step, S, cur ^ 1, 2);
    if (rc) return rc;
    return compute_flows(ctx, s, S, cur, d_flows, flow_frame_stride, flow_step, cur ^ 1, 2);
}

extern "C" int rcflow_push_batch_dev(rc_ctx* ctx, int stream, const uint8_t* d_frames, size_t frame_stride, size_t step,
                                     int nstreams, int w, int h, float* d_flows, size_t flow_frame_stride,
                                     size_t flow_step, const rc_farneback_params* p, int use_graph) {
    RcSlot* s = rc_slot(ctx, stream);
    if (!s || !d_frames || nstreams < 1 || nstreams > 4096) { if (s) rc_set_error("bad batch arguments"); return RC_EINVAL; }
    if (step < (size_t)w || (nstreams > 1 && frame_stride < step * (size_t)(h - 1) + w)) {
        rc_set_error("batch strides smaller than a frame");
        return RC_EINVAL;
    }
    RC_HIP(hipSetDevice(ctx->device));
    // every stream keeps a ring of two expansions: slot 2*z + parity
    int rc = ensure_plan(ctx, *s, w, h, p, nstreams, 2 * nstreams);
    if (rc) return rc;
    s->primed = 0;
    if (!s->batch_primed) {
        rc_batch_graph_drop(*s);
        if ((rc = expand_frames(ctx, *s, d_frames, frame_stride, step, nstreams, 0, 2))) return rc;
        s->batch_primed = 1;
        s->batch_cur = 0;
        return 1;
    }
    if (!d_flows || flow_step < (size_t)w * 8 ||
        (nstreams > 1 && flow_frame_stride < flow_step * (size_t)(h - 1) + (size_t)w * 8)) {
        rc_set_error("bad flow batch buffer");
        return RC_EINVAL;
    }
    const int cur = s->batch_cur;
    RcBatchKey key = {d_frames, frame_stride, step, d_flows, flow_frame_stride, flow_step, (void*)s->cur};
    const bool graph_ok = use_graph && !ctx->prof_on;
    if (graph_ok && s->batch_exec[cur] && !memcmp(&key, &s->batch_key[cur], sizeof(key))) {
        // steady state: replay the captured launch sequence of this parity
        RC_HIP(hipGraphLaunch((hipGraphExec_t)s->batch_exec[cur], s->cur));
    } else if (graph_ok && s->batch_eager[cur] && !memcmp(&key, &s->batch_key[cur], sizeof(key))) {
        // second time with the same buffers: capture while launching
        // the null stream cannot capture: record on the slot's own stream (capture executes
        // nothing) and launch the instantiated graph on the caller's stream
        hipGraph_t graph = nullptr;
        hipStream_t run_stream = s->cur, cap_stream = s->cur ? s->cur : s->own;
        RC_HIP(hipStreamBeginCapture(cap_stream, hipStreamCaptureModeRelaxed));
        s->cur = cap_stream;
        rc = batch_launches(ctx, *s, d_frames, frame_stride, step, nstreams, d_flows, flow_frame_stride, flow_step, cur);
        s->cur = run_stream;
        hipError_t e = hipStreamEndCapture(cap_stream, &graph);
        if (rc) { if (graph) (void)hipGraphDestroy(graph); return rc; }
        if (e != hipSuccess || !graph) { rc_set_error("hipStreamEndCapture failed: %s", hipGetErrorString(e)); return RC_EHIP; }
        hipGraphExec_t exec = nullptr;
        e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
        (void)hipGraphDestroy(graph);
        if (e != hipSuccess) { rc_set_error("hipGraphInstantiate failed: %s", hipGetErrorString(e)); return RC_EHIP; }
        if (s->batch_exec[cur]) (void)hipGraphExecDestroy((hipGraphExec_t)s->batch_exec[cur]);
        s->batch_exec[cur] = exec;
        RC_HIP(hipGraphLaunch(exec, s->cur));
    } else {
        if (s->batch_exec[cur]) { (void)hipGraphExecDestroy((hipGraphExec_t)s->batch_exec[cur]); s->batch_exec[cur] = nullptr; }
        if ((rc = batch_launches(ctx, *s, d_frames, frame_stride, step, nstreams, d_flows, flow_frame_stride, flow_step, cur)))
            return rc;
        s->batch_key[cur] = key;
        s->batch_eager[cur] = 1;
    }
    s->batch_cur = cur ^ 1;
    return RC_OK;
}

extern "C" int rcflow_batch_reset(rc_ctx* ctx, int stream) {
    RcSlot* s = rc_slot(ctx, stream);
    if (!s) return RC_EINVAL;
    RC_HIP(hipStreamSynchronize(s->cur));
    rc_batch_graph_drop(*s);
    s->batch_primed = 0;
    s->batch_cur = 0;
    return RC_OK;
}

// ---------------------------------------------------------------------------- stage entry points
extern "C" int rcflow_stage_pyr_level_dev(rc_ctx* ctx, int stream, const uint8_t* d_img, size_t step, int w,
                                          int h, double pyr_scale, int k, float* d_out) {
    RcSlot* s = rc_slot(ctx, stream);
    if (!s || !d_img || !d_out || k < 0 || k >= RC_MAX_LEVELS || !(pyr_scale > 0 && pyr_scale < 1)) return RC_EINVAL;
    RC_HIP(hipSetDevice(ctx->device));
    RcLevel L;
    level_geom(w, h, pyr_scale, k, L);
    pick_pyr_tile(L, w, h);
    std::vector<float> kh(L.ksize);
    host_gaussian_kernel(L.ksize, L.sigma > 0 ? L.sigma : 0., kh.data());
    int rc = rc_buf_ensure(s->stage_f32[0], kh.size() * sizeof(float));
    if (rc) return rc;
    RC_HIP(hipMemcpyAsync(s->stage_f32[0].p, kh.data(), kh.size() * sizeof(float), hipMemcpyHostToDevice, s->cur));
    RC_HIP(hipStreamSynchronize(s->cur));
    RcPyrArgs pa;
    pa.src = d_img; pa.src_step = step; pa.src_frame_stride = 0; pa.W0 = w; pa.H0 = h;
    pa.dst = d_out; pa.dst_slot_stride = 0; pa.dslot0 = 0; pa.nslots = 1; pa.zstep = 1;
    pa.w = L.w; pa.h = L.h; pa.scale_x = L.scale_x; pa.scale_y = L.scale_y;
    pa.ksize = L.ksize; pa.kern = (const float*)s->stage_f32[0].p;
    pa.tw = L.pyr_tw; pa.th = L.pyr_th; pa.reg_wp = L.pyr_reg_w; pa.reg_hmax = L.pyr_reg_h;
    pa.direct = (ctx->ablate & RC_ABL_PYR_STAGED) != 0;
    pa.fixed3 = k == 0 && L.ksize == 3 && !(L.sigma > 0) && L.w == w && L.h == h;
    rc_launch_pyr(pa, 1, L.pyr_lds, s->cur);
    RC_HIP(hipGetLastError());
    return RC_OK;
}

extern "C" int rcflow_stage_polyexp_dev(rc_ctx* ctx, int stream, const float* d_I, int w, int h, int poly_n,
                                        double poly_sigma, float* d_R5) {
    RcSlot* s = rc_slot(ctx, stream);
    if (!s || !d_I || !d_R5 || w <= 0 || h <= 0 || poly_n < 1 || poly_n > RC_MAX_POLY_N) return RC_EINVAL;
    RC_HIP(hipSetDevice(ctx->device));
    size_t n = (size_t)w * h;
    int rc;
    if ((rc = rc_buf_ensure(s->stage_f32[0], n * sizeof(float4)))) return rc;
    if ((rc = rc_buf_ensure(s->stage_f32[1], n * sizeof(float)))) return rc;
    RcPolyArgs qa;
    memset(&qa, 0, sizeof(qa));
    qa.I = d_I; qa.I_slot_stride = 0;
    qa.RA = (float4*)s->stage_f32[0].p; qa.RB = (float*)s->stage_f32[1].p; qa.R_slot_stride = 0;
    qa.slot0 = 0; qa.nslots = 1; qa.zstep = 1; qa.w = w; qa.h = h; qa.tile_h = ctx->poly_tile_h; qa.valu_vertical = !ctx->poly_mfma;
    if ((rc = host_prepare_poly(poly_n, poly_sigma, ctx->exact_taps || ctx->exact == 1, qa.pk))) return rc;
    if (ctx->exact == 1) rc_launch_exact_polyexp(qa, 1, s->cur);
    else rc_launch_polyexp(qa, 1, s->cur);
    rc_launch_unpack_R5(qa.RA, qa.RB, d_R5, (int)n, s->cur);
    RC_HIP(hipGetLastError());
    return RC_OK;
}

extern "C" int rcflow_stage_flow_iter_dev(rc_ctx* ctx, int stream, const float* d_R0, const float* d_R1,
                                          const float* d_flow_in, int w, int h, int winsize, int flags,
                                          float* d_flow_out) {
    RcSlot* s = rc_slot(ctx, stream);
    if (!s || !d_R0 || !d_R1 || !d_flow_out || w <= 0 || h <= 0 || winsize < 1 || winsize / 2 > 24 ||
        (flags & ~RC_FARNEBACK_GAUSSIAN))
        return RC_EINVAL;
    RC_HIP(hipSetDevice(ctx->device));
    size_t n = (size_t)w * h;
    int rc;
    if ((rc = rc_buf_ensure(s->stage_f32[0], 2 * n * sizeof(float4)))) return rc;
    if ((rc = rc_buf_ensure(s->stage_f32[1], 2 * n * sizeof(float)))) return rc;
    float4* RA = (float4*)s->stage_f32[0].p;
    float* RB = (float*)s->stage_f32[1].p;
    rc_launch_pack_R5(d_R0, RA, RB, (int)n, s->cur);
    rc_launch_pack_R5(d_R1, RA + n, RB + n, (int)n, s->cur);
    RcWindow win;
    host_window(winsize, flags, win);
    if (ctx->exact == 1) {
        if ((rc = rc_buf_ensure(s->exM, n * 5 * sizeof(float)))) return rc;
        if ((rc = rc_buf_ensure(s->exV, (size_t)w * ((h + 15) & ~15) * 5 * sizeof(double)))) return rc;
        if ((rc = rc_buf_ensure(s->exG, n * 5 * sizeof(double)))) return rc;
        if ((rc = rc_buf_ensure(s->stage_f32[2], n * sizeof(float2)))) return rc;
        RcExactArgs e;
        memset(&e, 0, sizeof(e));
        e.RA = RA; e.RB = RB; e.n = n; e.slot0 = 0; e.slot1 = 1; e.nslots = 2; e.zstep = 1; e.w = w; e.h = h; e.win = win; e.plain_scans = (ctx->ablate & RC_ABL_EXACT_PLAIN_SCANS) != 0;
        e.flow = (float2*)s->stage_f32[2].p; e.M = (float*)s->exM.p; e.V = s->exV.p; e.G = s->exG.p;
        if (d_flow_in) RC_HIP(hipMemcpyAsync(e.flow, d_flow_in, n * sizeof(float2), hipMemcpyDeviceToDevice, s->cur));
        else RC_HIP(hipMemsetAsync(e.flow, 0, n * sizeof(float2), s->cur));
        rc_launch_exact_matrices(e, 1, s->cur);
        e.out = (char*)d_flow_out; e.out_step = (size_t)w * 8; e.out_pair_stride = n * 8;
        rc_launch_exact_window_solve(e, 1, s->cur);
        RC_HIP(hipGetLastError());
        return RC_OK;
    }
    RcIterArgs a;
    memset(&a, 0, sizeof(a));
    a.RA = RA; a.RB = RB; a.R_slot_stride = n; a.slot0 = 0; a.slot1 = 1; a.nslots = 2; a.zstep = 1;
    a.w = w; a.h = h;
    a.in_mode = d_flow_in ? 1 : 0; a.fin = (const float2*)d_flow_in; a.fin_pair_stride = n;
    a.fout = (char*)d_flow_out; a.fout_step = (size_t)w * 8; a.fout_pair_stride = n * 8;
    a.solve = 1; a.win = win; a.xcd_remap = ctx->xcd_remap;
    rc_flow_fast::rc_launch_flow_iter(a, 1, s->cur);
    RC_HIP(hipGetLastError());
    return RC_OK;
}
