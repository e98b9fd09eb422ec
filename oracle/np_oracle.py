"""Independent float64 numpy/scipy restatement of the Farneback path (A1-A7).

TEST INFRASTRUCTURE, NOT PRODUCT.  PARITY UNPINNED (see rc_oracle.h).

Purpose: a second, differently structured statement of the same published algorithm
(OpenCV 4.1.0 modules/video/src/optflow.cpp; call site ripcurrents.cpp:215) used only to
cross-check oracle/farneback_oracle.cpp for transcription slips.  It is written with
whole-image dense operations (no row streaming, no stripe logic, float64 throughout), so
agreement with the C++ oracle to ~1e-4 also demonstrates SURVEY section 3.3's claim that
the in-loop stripe update of FarnebackUpdateFlow_* equals "blur+solve the whole image,
then update all matrices".
"""
import numpy as np
from scipy.ndimage import correlate1d


def cv_round(v):
    return int(np.rint(v))  # half to even


def level_geometry(w, h, pyr_scale, levels):
    k, scale = 0, 1.0
    while k < levels:
        scale *= pyr_scale
        if w * scale < 32 or h * scale < 32:
            break
        k += 1
    geoms = []
    for lv in range(k + 1):
        s = pyr_scale ** lv
        sigma = (1.0 / s - 1) * 0.5
        ks = max(cv_round(sigma * 5) | 1, 3)
        geoms.append(dict(w=cv_round(w * s), h=cv_round(h * s), sigma=sigma, ksize=ks))
    return geoms


def gaussian_kernel(n, sigma):
    small = {1: [1.0], 3: [0.25, 0.5, 0.25], 5: [0.0625, 0.25, 0.375, 0.25, 0.0625],
             7: [0.03125, 0.109375, 0.21875, 0.28125, 0.21875, 0.109375, 0.03125]}
    if n % 2 == 1 and n <= 7 and sigma <= 0:
        k = np.array(small[n], np.float64)
    else:
        s = sigma if sigma > 0 else ((n - 1) * 0.5 - 1) * 0.3 + 0.8
        x = np.arange(n) - (n - 1) * 0.5
        k = np.exp(-0.5 * x * x / (s * s))
    return k / k.sum()


def resize_linear(src, dw, dh):
    sh, sw = src.shape[:2]
    if (sw, sh) == (dw, dh):
        return src.copy()
    fx = (np.arange(dw) + 0.5) * (sw / dw) - 0.5
    fy = (np.arange(dh) + 0.5) * (sh / dh) - 0.5
    x0 = np.floor(fx).astype(int); ax = fx - x0
    y0 = np.floor(fy).astype(int); ay = fy - y0
    xa, xb = np.clip(x0, 0, sw - 1), np.clip(x0 + 1, 0, sw - 1)
    ya, yb = np.clip(y0, 0, sh - 1), np.clip(y0 + 1, 0, sh - 1)
    if src.ndim == 3:
        ax = ax[None, :, None]; ay = ay[:, None, None]
    else:
        ax = ax[None, :]; ay = ay[:, None]
    top = src[ya][:, xa] * (1 - ax) + src[ya][:, xb] * ax
    bot = src[yb][:, xa] * (1 - ax) + src[yb][:, xb] * ax
    return top * (1 - ay) + bot * ay


def pyr_level(img_u8, geom):
    f = img_u8.astype(np.float64)
    k = gaussian_kernel(geom["ksize"], geom["sigma"])
    f = correlate1d(f, k, axis=1, mode="mirror")   # BORDER_REFLECT_101
    f = correlate1d(f, k, axis=0, mode="mirror")
    return resize_linear(f, geom["w"], geom["h"])


def prepare_gaussian(n, sigma):
    if sigma < np.finfo(np.float32).eps:
        sigma = n * 0.3
    x = np.arange(-n, n + 1, dtype=np.float64)
    g = np.exp(-x * x / (2 * sigma * sigma))
    g /= g.sum()
    G = np.zeros((6, 6))
    gg = np.outer(g, g)                      # gg[y,x]
    X = x[None, :]; Y = x[:, None]
    G[0, 0] = gg.sum()
    G[1, 1] = (gg * X * X).sum()
    G[3, 3] = (gg * X ** 4).sum()
    G[5, 5] = (gg * X * X * Y * Y).sum()
    G[2, 2] = G[0, 3] = G[0, 4] = G[3, 0] = G[4, 0] = G[1, 1]
    G[4, 4] = G[3, 3]
    G[3, 4] = G[4, 3] = G[5, 5]
    inv = np.linalg.inv(G)
    return g, x * g, x * x * g, (inv[1, 1], inv[0, 3], inv[3, 3], inv[5, 5])


def polyexp(I, n, sigma):
    g, xg, xxg, (ig11, ig03, ig33, ig55) = prepare_gaussian(n, sigma)
    I = I.astype(np.float64)
    cv = lambda a, k, ax: correlate1d(a, k, axis=ax, mode="nearest")  # replicate border
    v0, v1, v2 = cv(I, g, 0), cv(I, xg, 0), cv(I, xxg, 0)
    b1, b2, b4 = cv(v0, g, 1), cv(v0, xg, 1), cv(v0, xxg, 1)
    b3, b6 = cv(v1, g, 1), cv(v1, xg, 1)
    b5 = cv(v2, g, 1)
    R = np.empty(I.shape + (5,))
    R[..., 0] = b3 * ig11
    R[..., 1] = b2 * ig11
    R[..., 2] = b1 * ig03 + b5 * ig33
    R[..., 3] = b1 * ig03 + b4 * ig33
    R[..., 4] = b6 * ig55
    return R


def update_matrices(R0, R1, flow):
    h, w = flow.shape[:2]
    ys, xs = np.meshgrid(np.arange(h), np.arange(w), indexing="ij")
    dx, dy = flow[..., 0], flow[..., 1]
    # float32 coordinate arithmetic, as the C++ does `float fx = x + dx`
    fx = (xs.astype(np.float32) + dx.astype(np.float32)).astype(np.float64)
    fy = (ys.astype(np.float32) + dy.astype(np.float32)).astype(np.float64)
    x1 = np.floor(fx).astype(int); y1 = np.floor(fy).astype(int)
    ax = (fx - x1)[..., None]; ay = (fy - y1)[..., None]
    inside = (x1 >= 0) & (x1 < w - 1) & (y1 >= 0) & (y1 < h - 1)
    xc, yc = np.clip(x1, 0, w - 2), np.clip(y1, 0, h - 2)
    samp = ((1 - ax) * (1 - ay) * R1[yc, xc] + ax * (1 - ay) * R1[yc, xc + 1]
            + (1 - ax) * ay * R1[yc + 1, xc] + ax * ay * R1[yc + 1, xc + 1])
    r2 = np.where(inside, samp[..., 0], 0.0)
    r3 = np.where(inside, samp[..., 1], 0.0)
    r4 = np.where(inside, (R0[..., 2] + samp[..., 2]) * 0.5, R0[..., 2])
    r5 = np.where(inside, (R0[..., 3] + samp[..., 3]) * 0.5, R0[..., 3])
    r6 = np.where(inside, (R0[..., 4] + samp[..., 4]) * 0.25, R0[..., 4] * 0.5)
    r2 = (R0[..., 0] - r2) * 0.5
    r3 = (R0[..., 1] - r3) * 0.5
    r2 = r2 + r4 * dy + r6 * dx
    r3 = r3 + r6 * dy + r5 * dx
    border = np.array([0.14, 0.14, 0.4472, 0.4472, 0.4472], np.float32).astype(np.float64)
    sx = np.ones(w); sy = np.ones(h)
    for i in range(min(5, w)):
        sx[i] *= border[i]; sx[w - 1 - i] *= border[i]
    for i in range(min(5, h)):
        sy[i] *= border[i]; sy[h - 1 - i] *= border[i]
    sc = sy[:, None] * sx[None, :]
    r2, r3, r4, r5, r6 = r2 * sc, r3 * sc, r4 * sc, r5 * sc, r6 * sc
    M = np.empty((h, w, 5))
    M[..., 0] = r4 * r4 + r6 * r6
    M[..., 1] = (r4 + r5) * r6
    M[..., 2] = r5 * r5 + r6 * r6
    M[..., 3] = r4 * r2 + r6 * r3
    M[..., 4] = r6 * r2 + r5 * r3
    return M


def blur_solve(M, winsize, gaussian):
    m = winsize // 2
    if gaussian:
        sigma = m * 0.3
        i = np.arange(-m, m + 1, dtype=np.float64)
        k = np.exp(-i * i / (2 * sigma * sigma)) if m > 0 else np.ones(1)
        k /= k.sum()
    else:
        k = np.ones(2 * m + 1) / (winsize * winsize) ** 0.5  # separable 1/(bs*bs) overall
    B = correlate1d(correlate1d(M, k, axis=0, mode="nearest"), k, axis=1, mode="nearest")
    g11, g12, g22, h1, h2 = [B[..., c] for c in range(5)]
    idet = 1.0 / (g11 * g22 - g12 * g12 + 1e-3)
    flow = np.empty(M.shape[:2] + (2,))
    flow[..., 0] = (g11 * h2 - g12 * h1) * idet
    flow[..., 1] = (g22 * h1 - g12 * h2) * idet
    return flow


def farneback(prev, nxt, pyr_scale=0.5, levels=2, winsize=3, iters=2, poly_n=15,
              poly_sigma=1.2, flags=0):
    h, w = prev.shape
    geoms = level_geometry(w, h, pyr_scale, levels)
    flow = None
    for geom in reversed(geoms):
        if flow is None:
            flow = np.zeros((geom["h"], geom["w"], 2))
        else:
            flow = resize_linear(flow, geom["w"], geom["h"]) * (1.0 / pyr_scale)
        R0 = polyexp(pyr_level(prev, geom), poly_n, poly_sigma)
        R1 = polyexp(pyr_level(nxt, geom), poly_n, poly_sigma)
        M = update_matrices(R0, R1, flow)
        for i in range(iters):
            flow = blur_solve(M, winsize, bool(flags & 256))
            if i < iters - 1:
                M = update_matrices(R0, R1, flow)
    return flow
