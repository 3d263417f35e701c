"""Seeded synthetic laser-grid-on-cylinder stereo frames (torch; runs on CPU or on the GPU).

The reference ships no data, so every benchmark / parity input is rendered
here: a calibrated stereo pair (K1, K2, T_C2_C1 as in utils/getCamParams.m:6-9) looks at a
cylinder of radius 45 mm (exp_gridDetection.m:39) on which a laser projector throws two fans of
planes (the grid) plus a saturated zero-order spot.  Frames come out undistorted, u8 grey, which is
what makePyGridPts.m:26 hands to detect_grid.  Ground truth (grid intersections with their
(col,row) indices in both images, the 3-D points and the axis) is returned with each batch.

Pixel constants of the reference are absolute (20-px opening kernels, 15-px windows, 10..5000 px^2
blobs), so the grid pitch is specified in pixels and the number of visible lines follows from the
image size.
"""
from dataclasses import dataclass, field
import math

import numpy as np
import torch


@dataclass
class Scene:
    h: int = 1200
    w: int = 1920
    radius: float = 45.0            # mm
    focal: float = None             # px; default 1.55*w (narrow lens, cylinder fills ~45% of the width)
    baseline: float = 90.0          # mm
    depth: tuple = (330.0, 400.0)   # mm, axis distance
    tilt_deg: float = 8.0           # axis tilt about camera z and x, uniform +-
    pitch_px: float = 34.0          # grid pitch in the image centre
    half_lines: int = None          # vertical grid lines -N..N (across the cylinder); default from the image size
    half_lines_v: int = None        # horizontal grid lines -Nv..Nv (along the axis)
    line_sigma: float = 1.6         # px
    peak: tuple = (185.0, 230.0)
    background: tuple = (6.0, 16.0)
    noise_sigma: float = 1.5
    spot_radius_px: tuple = (16.0, 26.0)

    def __post_init__(self):
        if self.focal is None:
            self.focal = 1.55 * self.w
        if self.half_lines is None:
            self.half_lines = max(3, int(0.32 * min(self.w * 0.55, self.h) / self.pitch_px))
        if self.half_lines_v is None:
            self.half_lines_v = max(3, int(0.43 * self.h / self.pitch_px))


def _rot(ax, ay, az):
    cx, sx, cy, sy, cz, sz = math.cos(ax), math.sin(ax), math.cos(ay), math.sin(ay), math.cos(az), math.sin(az)
    Rx = np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]])
    Ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
    Rz = np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]])
    return Rz @ Ry @ Rx


def make_rig(scene: Scene):
    """K1, K2 (3x3), T_C2_C1 (4x4: camera-1 coords -> camera-2 coords), projector pose (4x4, cam1 -> proj)."""
    f, w, h = scene.focal, scene.w, scene.h
    K1 = np.array([[f, 0, w / 2 - 0.5 + 3.0], [0, f, h / 2 - 0.5 - 2.0], [0, 0, 1.0]])
    K2 = np.array([[f * 1.004, 0, w / 2 - 0.5 - 4.0], [0, f * 1.004, h / 2 - 0.5 + 1.5], [0, 0, 1.0]])
    R21 = _rot(math.radians(0.4), math.radians(9.0), math.radians(-0.3))   # slight toe-in
    C2 = np.array([scene.baseline, 1.0, 3.0])                               # camera-2 centre in cam-1 coords
    T21 = np.eye(4)
    T21[:3, :3] = R21
    T21[:3, 3] = -R21 @ C2
    Rp = _rot(math.radians(-0.5), math.radians(4.5), math.radians(0.2))
    Cp = np.array([scene.baseline * 0.5, -18.0, 0.0])                       # projector between the cameras
    Tp = np.eye(4)
    Tp[:3, :3] = Rp
    Tp[:3, 3] = -Rp @ Cp
    return K1, K2, T21, Tp


def _frame_params(scene: Scene, n, seed):
    rng = np.random.default_rng(seed)
    depth = rng.uniform(*scene.depth, n)
    az = np.radians(rng.uniform(-scene.tilt_deg, scene.tilt_deg, n))
    ax = np.radians(rng.uniform(-scene.tilt_deg, scene.tilt_deg, n))
    dirs = np.stack([np.sin(az), np.cos(az) * np.cos(ax), np.sin(ax) * np.cos(az)], 1)
    dirs /= np.linalg.norm(dirs, axis=1, keepdims=True)
    org = np.stack([rng.uniform(-12, 12, n) + scene.baseline * 0.3, rng.uniform(-10, 10, n), depth], 1)
    peak = rng.uniform(*scene.peak, n)
    bg = rng.uniform(*scene.background, n)
    spot = rng.uniform(*scene.spot_radius_px, n)
    return dict(org=org, dir=dirs, peak=peak, bg=bg, spot=spot)


def _cyl_hit(o, d, c, a, R):
    """nearest intersection parameter of rays o + s d with the cylinder (c, a unit, R); torch, broadcast."""
    wv = o - c
    da = (d * a).sum(-1, keepdim=True)
    wa = (wv * a).sum(-1, keepdim=True)
    dp = d - da * a
    wp = wv - wa * a
    A = (dp * dp).sum(-1)
    B = 2 * (wp * dp).sum(-1)
    Cc = (wp * wp).sum(-1) - R * R
    disc = B * B - 4 * A * Cc
    hit = disc > 0
    s = (-B - torch.sqrt(disc.clamp_min(0))) / (2 * A)
    return s, hit & (s > 0)


def _render_view(scene, K, Tcam, Tp, fp, delta, device, gen, chunk=None):
    """render one view for all frames (batched over `chunk` frames per torch op). Tcam: cam1 -> this camera (4x4).
    returns u8 [n,h,w]"""
    n = fp['org'].shape[0]
    h, w = scene.h, scene.w
    dt = torch.float32
    if chunk is None:
        chunk = 8 if torch.device(device).type == 'cuda' else 1
    Kinv = torch.tensor(np.linalg.inv(K), dtype=dt, device=device)
    Rc = torch.tensor(Tcam[:3, :3], dtype=dt, device=device)
    tc = torch.tensor(Tcam[:3, 3], dtype=dt, device=device)
    Rp = torch.tensor(Tp[:3, :3], dtype=dt, device=device)
    tp = torch.tensor(Tp[:3, 3], dtype=dt, device=device)
    ys, xs = torch.meshgrid(torch.arange(h, device=device, dtype=dt), torch.arange(w, device=device, dtype=dt),
                            indexing='ij')
    pix = torch.stack([xs, ys, torch.ones_like(xs)], -1)           # [h,w,3]
    d_cam = pix @ Kinv.T                                            # ray dirs in this camera
    d1 = (d_cam @ Rc).unsqueeze(0)                                  # -> cam-1 coords, [1,h,w,3]
    o1 = -(Rc.T @ tc)                                               # camera centre in cam-1 coords
    Op = -(Rp.T @ tp)                                               # projector centre in cam-1 coords
    out = torch.empty((n, h, w), dtype=torch.uint8, device=device)
    N = scene.half_lines
    Nv = scene.half_lines_v
    for i0 in range(0, n, chunk):
        i1 = min(n, i0 + chunk)
        B = i1 - i0
        c = torch.tensor(fp['org'][i0:i1], dtype=dt, device=device).view(B, 1, 1, 3)
        a = torch.tensor(fp['dir'][i0:i1], dtype=dt, device=device).view(B, 1, 1, 3)
        peak = torch.tensor(fp['peak'][i0:i1], dtype=dt, device=device).view(B, 1, 1)
        bg = torch.tensor(fp['bg'][i0:i1], dtype=dt, device=device).view(B, 1, 1)
        spot_r = torch.tensor(fp['spot'][i0:i1], dtype=dt, device=device).view(B, 1, 1)
        s, hit = _cyl_hit(o1, d1, c, a, scene.radius)
        X = o1 + s.unsqueeze(-1) * d1                                # surface point, cam-1 coords [B,h,w,3]
        wv = X - c
        nrm = (wv - (wv * a).sum(-1, keepdim=True) * a) / scene.radius
        toP = Op - X
        ndot = (nrm * toP).sum(-1)
        lit = hit & (ndot > 0)
        cosi = (ndot / toP.norm(dim=-1)).clamp(0, 1)
        Xp = X @ Rp.T + tp
        pa = Xp[..., 0] / Xp[..., 2] / delta                         # grid coordinates (integers on lines)
        pb = Xp[..., 1] / Xp[..., 2] / delta
        del X, wv, nrm, toP, Xp
        img = bg.expand(B, h, w).clone()
        amp = peak * (0.45 + 0.55 * cosi)
        jac = []
        for q, other, nq, no in ((pa, pb, N, Nv), (pb, pa, Nv, N)):
            gy_, gx_ = torch.gradient(q, dim=(1, 2))
            gnorm = torch.sqrt(gx_ * gx_ + gy_ * gy_).clamp_min(1e-6)
            jac.append(gnorm)
            r = torch.round(q)
            dist = (q - r).abs() / gnorm                             # px distance to the nearest grid line
            on = lit & (r.abs() <= nq) & (other.abs() <= no + 0.35) & (gnorm < 0.5)
            img = img + torch.where(on, amp * torch.exp(-0.5 * (dist / scene.line_sigma) ** 2), torch.zeros_like(img))
        # zero-order spot at (0,0)
        rr = torch.sqrt((pa / jac[0]) ** 2 + (pb / jac[1]) ** 2)
        img = img + torch.where(lit, 400.0 * torch.sigmoid((spot_r - rr) * 1.2), torch.zeros_like(img))
        for k in range(B):
            img[k] += scene.noise_sigma * torch.randn((h, w), generator=gen, device=device, dtype=dt)
        out[i0:i1] = img.round().clamp(0, 255).to(torch.uint8)
    return out


def ground_truth(scene: Scene, K1, K2, T21, Tp, fp, delta):
    """per frame: dict(idx [m,2] (col,row) projector indices, X [m,3], uv1 [m,2], uv2 [m,2])"""
    N, Nv = scene.half_lines, scene.half_lines_v
    ii, jj = np.meshgrid(np.arange(-N, N + 1), np.arange(-Nv, Nv + 1), indexing='ij')
    rays_p = np.stack([ii.ravel() * delta, jj.ravel() * delta, np.ones(ii.size)], 1)
    Rp, tp = Tp[:3, :3], Tp[:3, 3]
    Op = -Rp.T @ tp
    d = rays_p @ Rp                                                  # Rp^T applied
    res = []
    for i in range(fp['org'].shape[0]):
        c, a = fp['org'][i], fp['dir'][i]
        wv = Op - c
        da = d @ a
        dp = d - da[:, None] * a
        wp = wv - (wv @ a) * a
        A = (dp * dp).sum(1); B = 2 * dp @ wp; Cc = wp @ wp - scene.radius ** 2
        disc = B * B - 4 * A * Cc
        ok = disc > 0
        s = (-B - np.sqrt(np.maximum(disc, 0))) / (2 * A)
        X = Op + s[:, None] * d
        X2 = X @ T21[:3, :3].T + T21[:3, 3]
        p1 = X @ K1.T; uv1 = p1[:, :2] / p1[:, 2:3]
        p2 = X2 @ K2.T; uv2 = p2[:, :2] / p2[:, 2:3]
        # visible from both cameras (front-facing)
        nrm = (X - c) - ((X - c) @ a)[:, None] * a
        v1 = (nrm * (-X)).sum(1) > 0
        C2 = -T21[:3, :3].T @ T21[:3, 3]
        v2 = (nrm * (C2 - X)).sum(1) > 0
        inimg = (uv1[:, 0] > 0) & (uv1[:, 0] < scene.w - 1) & (uv1[:, 1] > 0) & (uv1[:, 1] < scene.h - 1) & \
                (uv2[:, 0] > 0) & (uv2[:, 0] < scene.w - 1) & (uv2[:, 1] > 0) & (uv2[:, 1] < scene.h - 1)
        m = ok & v1 & v2 & inimg
        res.append(dict(idx=np.stack([ii.ravel(), jj.ravel()], 1)[m], X=X[m], uv1=uv1[m], uv2=uv2[m]))
    return res


def render_batch(n, h=1200, w=1920, seed=0, device='cpu', scene: Scene = None, with_gt=True):
    """n stereo frames.  returns dict(left u8[n,h,w], right u8[n,h,w], K1, K2, T21, radius, axis_org, axis_dir, gt)"""
    scene = scene or Scene(h=h, w=w)
    K1, K2, T21, Tp = make_rig(scene)
    fp = _frame_params(scene, n, seed)
    delta = scene.pitch_px / scene.focal * 1.0                      # projector angular pitch (tan units)
    gen = torch.Generator(device=device)
    gen.manual_seed(1000003 * seed + 17)
    left = _render_view(scene, K1, np.eye(4), Tp, fp, delta, device, gen)
    right = _render_view(scene, K2, T21, Tp, fp, delta, device, gen)
    out = dict(left=left, right=right, K1=K1, K2=K2, T21=T21, radius=scene.radius,
               axis_org=fp['org'], axis_dir=fp['dir'], scene=scene)
    if with_gt:
        out['gt'] = ground_truth(scene, K1, K2, T21, Tp, fp, delta)
    return out


def degrade(img, rng):
    """random degradations of a rendered frame (tools/stress_parity.py, tests): exposure, uniform sensor noise, dead bands,
    dark / saturated boxes, an intensity ramp.  Returns (u8 image, list of what was applied); the order of the draws from
    `rng` is part of the contract (a seed names a frame)."""
    f = img.astype(np.float64)
    h, w = f.shape
    what = []
    if rng.random() < 0.6:
        s = rng.uniform(0.45, 1.0); f *= s; what.append(f'exposure {s:.2f}')
    if rng.random() < 0.5:
        a = int(rng.integers(1, 12)); f += rng.integers(-a, a + 1, size=f.shape); what.append(f'noise {a}')
    if rng.random() < 0.3:
        x = int(rng.integers(w // 4, 3 * w // 4)); k = int(rng.integers(2, 9)); f[:, x:x + k] = 0; what.append('v band')
    if rng.random() < 0.3:
        y = int(rng.integers(h // 4, 3 * h // 4)); k = int(rng.integers(2, 9)); f[y:y + k, :] = 0; what.append('h band')
    for _ in range(int(rng.integers(0, 4))):
        y, x = int(rng.integers(0, h - 40)), int(rng.integers(0, w - 40))
        hh, ww = int(rng.integers(8, 40)), int(rng.integers(8, 40))
        v = 255 if rng.random() < 0.4 else int(rng.integers(0, 60))
        f[y:y + hh, x:x + ww] = v; what.append(f'box {v}')
    if rng.random() < 0.2:
        f += np.linspace(0, rng.uniform(20, 80), w)[None, :]; what.append('gradient')
    return np.clip(f, 0, 255).astype(np.uint8), what
