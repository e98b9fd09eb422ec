"""Deterministic synthetic clips.

The reference ships no sample video (inputs are user-supplied, ripcurrents.cpp:61-66),
so every test and bench input is generated here: 8-bit gray frames as the frame loop
hands them to calcOpticalFlowFarneback after resize+cvtColor (ripcurrents.cpp:209-213).

Works on numpy arrays, or on torch tensors when `device` is given (bench: frames are
generated directly in HBM).
"""
import math

import numpy as np

N_WAVES = 64


def _texture_params(seed):
    rng = np.random.RandomState(seed)
    # band-limited 1/f texture: wavelengths 5..80 px, random orientation and phase
    wavelength = np.exp(rng.uniform(math.log(5.0), math.log(80.0), N_WAVES))
    theta = rng.uniform(0, 2 * math.pi, N_WAVES)
    phase = rng.uniform(0, 2 * math.pi, N_WAVES)
    freq = 1.0 / wavelength
    amp = wavelength / wavelength.max()          # ~1/f
    amp = amp / math.sqrt((amp ** 2).sum() / 2)  # unit variance
    kx = 2 * math.pi * freq * np.cos(theta)
    ky = 2 * math.pi * freq * np.sin(theta)
    return kx, ky, phase, amp


def _eval_texture(xs, ys, params, xp, chunk=8):
    """sum_i amp_i sin(kx_i x + ky_i y + phase_i) at per-pixel coordinates xs, ys."""
    kx, ky, ph, amp = params
    out = xp.zeros_like(xs)
    for i in range(N_WAVES):
        out = out + float(amp[i]) * xp.sin(float(kx[i]) * xs + float(ky[i]) * ys + float(ph[i]))
    return out


def _grid(w, h, xp, device, dtype):
    if device is None:
        ys, xs = np.meshgrid(np.arange(h, dtype=dtype), np.arange(w, dtype=dtype), indexing="ij")
    else:
        import torch
        tdt = torch.float64 if dtype == np.float64 else torch.float32
        ys, xs = torch.meshgrid(torch.arange(h, dtype=tdt, device=device),
                                torch.arange(w, dtype=tdt, device=device), indexing="ij")
    return xs, ys


def _quantise(img, xp, device):
    v = xp.clip(xp.round(128.0 + 40.0 * img), 0, 255)
    if device is None:
        return v.astype(np.uint8)
    import torch
    return v.to(torch.uint8)


def _backend(device):
    if device is None:
        return np
    import torch
    return torch


def translating_clip(w=640, h=480, frames=2, u=1.25, v=-0.75, seed=1234, device=None):
    """Config C1: texture translated by a constant (u,v) px/frame; true flow = (u,v)."""
    xp = _backend(device)
    params = _texture_params(seed)
    xs, ys = _grid(w, h, xp, device, np.float64)
    out = []
    for t in range(frames):
        out.append(_quantise(_eval_texture(xs - u * t, ys - v * t, params, xp), xp, device))
    return xp.stack(out)


def surf_field(w, h, xp=np, device=None, dtype=np.float64):
    """Velocity field of the synthetic surf clip: shoreward drift (0,+1.5) px/frame plus a
    Gaussian-profile seaward jet (the rip) whose centre-line speed is -2.0 px/frame."""
    xs, ys = _grid(w, h, xp, device, dtype)
    jet = xp.exp(-((xs - 0.55 * w) ** 2) / (2.0 * (0.06 * w) ** 2))
    U = 0.25 * jet * xp.sin(2 * math.pi * ys / (0.5 * h))
    V = 1.5 - 3.5 * jet
    return U, V


def surf_clip(w=1920, h=1080, frames=2, seed=1234, device=None, t0=0):
    """Config C2/C3 'synthetic surf': texture advected by surf_field plus travelling
    sinusoidal wave fronts.  Displacement is reset every 8 frames so shear stays bounded."""
    xp = _backend(device)
    dtype = np.float64 if device is None else np.float32
    params = _texture_params(seed)
    xs, ys = _grid(w, h, xp, device, dtype)
    U, V = surf_field(w, h, xp, device, dtype)
    out = []
    for t in range(t0, t0 + frames):
        tt = float(t % 8) + 8.0 * 0.37 * (t // 8)   # piecewise clip time, deterministic
        tex = _eval_texture(xs - U * tt, ys - V * tt, params, xp)
        fronts = 0.6 * xp.sin(2 * math.pi * (ys - 2.5 * t) / (0.11 * h) + 0.002 * xs)
        out.append(_quantise(0.85 * tex + fronts, xp, device))
    return xp.stack(out)


def rotation_field(w=640, h=480):
    """The analytic rotational field of validate_streamlines (main.cpp:372-380)."""
    rows, cols = np.meshgrid(np.arange(h), np.arange(w), indexing="ij")
    flow = np.zeros((h, w, 2), np.float32)
    flow[..., 0] = (-(rows - h / 2.0) / h * 100).astype(np.float32)
    flow[..., 1] = ((cols - w / 2.0) / w * 100).astype(np.float32)
    return flow
