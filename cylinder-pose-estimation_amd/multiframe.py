"""Row f-1: the multi-frame camera<->AGV fit, utils/fitCylinderWPts3sAngs.m (+ getTAGVcyl.m, vec2T.m,
T2vec.m), the immediate consumer of the per-frame outputs (exp_gridDetection.m:87).

One 6-parameter Nelder-Mead (MATLAB fminsearch order) over  v(agvPose) = sum_i mean((d_i - R)^2).  The data-parallel
part -- distances of every frame's points to that frame's predicted axis -- is the HIP kernel behind
cpe_multi_frame_terms (one wavefront per frame, points stay in HBM); the simplex logic is host code in plain Python
floats (IEEE double, libm sin/cos/acos), which is what MATLAB's interpreter does for the reference.

[ext] rotvec2mat3d / rotmat2vec3d / mrdivide / fminsearch are restated as in oracle/src/orc_fit.c (parity unpinned).
Reproduced quirk: cylParams{i} is the 2x6 [cylParams0; cylParams] matrix and applyCylParamsPrior indexes it linearly
(applyCylParamsPrior.m:6-7), so the "origin"/"direction" used for the initial pose mix the two rows.
"""
import math

import torch

from . import lib as _lib


def get_TAGVcyl(pan, tilt):
    """getTAGVcyl.m (default config), row-major 4x4 as a flat list"""
    cp, sp, ct, st = math.cos(pan), math.sin(pan), math.cos(-tilt), math.sin(-tilt)
    TAP = [cp, -sp, 0, 0, sp, cp, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1]
    TPT0 = [1, 0, 0, -143.1, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1]
    L = math.sqrt((-143.1 * -143.1 + 0.0 * 0.0) + 0.0 * 0.0)
    mtr = -math.tan(tilt) * L
    T01 = [1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, mtr, 0, 0, 0, 1]
    T12 = [ct, 0, st, 0, 0, 1, 0, 0, -st, 0, ct, 0, 0, 0, 0, 1]
    T2C = [0, -1, 0, 321.1, -1, 0, 0, 0, 0, 0, -1, 110, 0, 0, 0, 1]
    acc = [float(v) for v in TAP]
    for M in (TPT0, T01, T12, T2C):
        nxt = [0.0] * 16
        for r in range(4):
            for c in range(4):
                s = 0.0
                for k in range(4):
                    s = s + acc[r * 4 + k] * M[k * 4 + c]
                nxt[r * 4 + c] = s
        acc = nxt
    return acc


def _rotvec2mat(v):
    th = math.sqrt((v[0] * v[0] + v[1] * v[1]) + v[2] * v[2])
    if th < 1e-6:
        return [1.0, 0, 0, 0, 1.0, 0, 0, 0, 1.0]
    u = [v[0] / th, v[1] / th, v[2] / th]
    c, s, t = math.cos(th), math.sin(th), 1 - math.cos(th)
    K = [0, -u[2], u[1], u[2], 0, -u[0], -u[1], u[0], 0]
    return [(c * (1.0 if r == q else 0.0) + t * (u[r] * u[q])) + s * K[r * 3 + q] for r in range(3) for q in range(3)]


def _mat2rotvec(R):
    t = (R[0] + R[4]) + R[8]
    ca = min(1.0, max(-1.0, (t - 1) / 2))
    th = math.acos(ca)
    r = [R[7] - R[5], R[2] - R[6], R[3] - R[1]]
    if math.sin(th) >= 1e-4:
        vth = 1 / (2 * math.sin(th))
        return [th * (r[k] * vth) for k in range(3)]
    if t - 1 > 0:
        return [(.5 - (t - 3) / 12) * r[k] for k in range(3)]
    a = 0
    if R[4] > R[a * 4]:
        a = 1
    if R[8] > R[a * 4]:
        a = 2
    b, c = (a + 1) % 3, (a + 2) % 3
    s = math.sqrt(R[a * 4] - R[b * 4] - R[c * 4] + 1)
    w = [0.0, 0.0, 0.0]
    w[a] = s / 2
    w[b] = (R[b * 3 + a] + R[a * 3 + b]) / (2 * s)
    w[c] = (R[c * 3 + a] + R[a * 3 + c]) / (2 * s)
    nw = math.sqrt((w[0] * w[0] + w[1] * w[1]) + w[2] * w[2])
    return [th * w[k] / nw for k in range(3)]


def vec2T(x):
    """vec2T.m -> flat row-major 4x4"""
    R = _rotvec2mat(x)
    T = [0.0] * 16
    for r in range(3):
        for c in range(3):
            T[r * 4 + c] = R[r * 3 + c]
        T[r * 4 + 3] = x[3 + r]
    T[15] = 1.0
    return T


def T2vec(T):
    R = [T[r * 4 + c] for r in range(3) for c in range(3)]
    return _mat2rotvec(R) + [T[3], T[7], T[11]]


def _cross(a, b):
    return [a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]]


def _apply_prior(cyl, ymin):
    o, d = list(cyl[:3]), list(cyl[3:])
    if d[1] < 0:
        d = [-d[0], -d[1], -d[2]]
    t = 0.0
    if not (abs(d[1]) < 2.220446049250313e-16):
        t = (ymin - o[1]) / d[1]
    return [o[c] + t * d[c] for c in range(3)] + d


def initial_pose(cyl_raw01, ymin01, TAGV01):
    """T0 of fitCylinderWPts3sAngs.m:40-69 as a rotation-vector pose (6 floats)"""
    cp = []
    for i in range(2):
        M = cyl_raw01[i]                                  # 2 x 6 nested list [cylParams0; cylParams]
        lin = [M[k % 2][k // 2] for k in range(6)]        # MATLAB linear (column-major) indexing of the 2x6 matrix
        cp.append(_apply_prior(lin, ymin01[i]))
    A1, A2 = TAGV01
    p1, p2 = [A1[3], A1[7], A1[11]], [A2[3], A2[7], A2[11]]
    y1 = [A1[1], A1[5], A1[9]]
    d12 = [p2[k] - p1[k] for k in range(3)]
    nd = _cross(y1, d12)
    nn = math.sqrt((nd[0] * nd[0] + nd[1] * nd[1]) + nd[2] * nd[2])
    nd = [v / nn for v in nd]
    ed12 = [cp[1][k] - cp[0][k] for k in range(3)]
    dir1 = cp[0][3:]
    en = _cross(dir1, ed12)
    ne = math.sqrt((en[0] * en[0] + en[1] * en[1]) + en[2] * en[2])
    en = [v / ne for v in en]
    c1, c2 = _cross(dir1, en), _cross(y1, nd)
    Ac, Bc = [dir1, en, c1], [y1, nd, c2]
    Bt = [Bc[r][c] for r in range(3) for c in range(3)]
    At = [Ac[r][c] for r in range(3) for c in range(3)]
    for c in range(3):                                    # R = A / B: Gaussian elimination on B' R' = A'
        pv = c
        for r in range(c + 1, 3):
            if abs(Bt[r * 3 + c]) > abs(Bt[pv * 3 + c]):
                pv = r
        if pv != c:
            for k in range(3):
                Bt[c * 3 + k], Bt[pv * 3 + k] = Bt[pv * 3 + k], Bt[c * 3 + k]
                At[c * 3 + k], At[pv * 3 + k] = At[pv * 3 + k], At[c * 3 + k]
        for r in range(c + 1, 3):
            f = Bt[r * 3 + c] / Bt[c * 3 + c]
            for k in range(c, 3):
                Bt[r * 3 + k] = Bt[r * 3 + k] - f * Bt[c * 3 + k]
            for k in range(3):
                At[r * 3 + k] = At[r * 3 + k] - f * At[c * 3 + k]
    Rt = [0.0] * 9
    for k in range(3):
        for r in (2, 1, 0):
            s = At[r * 3 + k]
            for q in range(r + 1, 3):
                s = s - Bt[r * 3 + q] * Rt[q * 3 + k]
            Rt[r * 3 + k] = s / Bt[r * 3 + r]
    T0 = [0.0] * 16
    for r in range(3):
        for c in range(3):
            T0[r * 4 + c] = Rt[c * 3 + r]
    for r in range(3):
        s = (T0[r * 4] * p1[0] + T0[r * 4 + 1] * p1[1]) + T0[r * 4 + 2] * p1[2]
        T0[r * 4 + 3] = cp[0][r] - s
    T0[15] = 1.0
    return T2vec(T0)


def _eps(x):
    x = abs(x)
    if x < 2.2250738585072014e-308:
        return 5e-324
    m, e = math.frexp(x)
    return math.ldexp(1.0, e - 53)


def nelder_mead6(fobj, x0, tolx=1e-5, tolf=1e-5, maxiter=100000, maxfun=100000):
    """MATLAB fminsearch.m, n = 6 (same order of operations as oracle/src/orc_fit.c::nelder_mead6_fn)"""
    N = 6
    v = [list(x0)]
    fv = [fobj(v[0])]
    for j in range(N):
        y = list(x0)
        y[j] = (1 + 0.05) * y[j] if y[j] != 0 else 0.00025
        v.append(y)
        fv.append(fobj(y))

    def sort_simplex():
        order = sorted(range(N + 1), key=lambda k: fv[k])     # Python's sort is stable, as MATLAB's
        return [v[k] for k in order], [fv[k] for k in order]

    v, fv = sort_simplex()
    func_evals, itercount = N + 1, 1
    while func_evals < maxfun and itercount < maxiter:
        df = max(abs(fv[0] - fv[j]) for j in range(1, N + 1))
        dx = max(abs(v[j][k] - v[0][k]) for j in range(1, N + 1) for k in range(N))
        if df <= max(tolf, 10 * _eps(fv[0])) and dx <= max(tolx, 10 * _eps(max(v[0]))):
            break
        xbar = []
        for k in range(N):
            s = v[0][k]
            for j in range(1, N):
                s = s + v[j][k]
            xbar.append(s / N)
        xr = [2.0 * xbar[k] - 1.0 * v[N][k] for k in range(N)]
        fxr = fobj(xr); func_evals += 1
        shrink = False
        if fxr < fv[0]:
            xe = [3.0 * xbar[k] - 2.0 * v[N][k] for k in range(N)]
            fxe = fobj(xe); func_evals += 1
            if fxe < fxr:
                v[N], fv[N] = xe, fxe
            else:
                v[N], fv[N] = xr, fxr
        elif fxr < fv[N - 1]:
            v[N], fv[N] = xr, fxr
        elif fxr < fv[N]:
            xc = [1.5 * xbar[k] - 0.5 * v[N][k] for k in range(N)]
            fxc = fobj(xc); func_evals += 1
            if fxc <= fxr:
                v[N], fv[N] = xc, fxc
            else:
                shrink = True
        else:
            xcc = [0.5 * xbar[k] + 0.5 * v[N][k] for k in range(N)]
            fxcc = fobj(xcc); func_evals += 1
            if fxcc < fv[N]:
                v[N], fv[N] = xcc, fxcc
            else:
                shrink = True
        if shrink:
            for j in range(1, N + 1):
                v[j] = [v[0][k] + 0.5 * (v[j][k] - v[0][k]) for k in range(N)]
                fv[j] = fobj(v[j])
            func_evals += N
        v, fv = sort_simplex()
        itercount += 1
    return v[0], fv[0], itercount, func_evals


class MultiFrameObjective:
    """v(agvPose) with the per-frame terms computed on the GPU"""

    def __init__(self, pts3, cnt, TAGV, radius):
        self.L = _lib.load()
        self.pts3, self.cnt, self.radius = pts3.contiguous(), cnt.contiguous(), float(radius)
        self.F = cnt.shape[0]
        self.TAGV = torch.tensor(TAGV, dtype=torch.float64, device=pts3.device).reshape(self.F, 16).contiguous()
        self.terms = torch.zeros(self.F, dtype=torch.float64, device=pts3.device)
        self.Tdev = torch.zeros(16, dtype=torch.float64, device=pts3.device)

    def __call__(self, x):
        self.Tdev.copy_(torch.tensor(vec2T(x), dtype=torch.float64))
        _lib.check(self.L.cpe_multi_frame_terms(self.pts3.data_ptr(), self.cnt.data_ptr(), self.F, self.TAGV.data_ptr(),
                                                self.Tdev.data_ptr(), self.radius, self.terms.data_ptr(),
                                                torch.cuda.current_stream().cuda_stream), 'cpe_multi_frame_terms')
        v = 0.0
        for t in self.terms.tolist():                      # v = v + (vi*vi')/length(vi), in frame order
            v = v + t
        return v


def fit_multi_frame(pts3, cnt, cyl_raw, angles, radius):
    """[T, fval] = fitCylinderWPts3sAngs(Pts3s, angs, cylRadius)
    pts3 f64[F,MAXP,3] / cnt i32[F] / cyl_raw f64[F,2,6] as returned by fit_cylinder_batch (fitCylinderWPts3 per frame),
    angles: F x 2 (pan, tilt) in rad  ->  dict(T 4x4 row-major list, x, x0, fvals [f0, f], iters, evals)"""
    F = cnt.shape[0]
    assert F >= 2 and len(angles) == F
    TAGV = [get_TAGVcyl(float(a[0]), float(a[1])) for a in angles]
    raw01 = cyl_raw[:2].cpu().tolist()
    ymin01 = [float(pts3[i, :int(cnt[i]), 1].min()) for i in range(2)]
    x0 = initial_pose(raw01, ymin01, TAGV[:2])
    obj = MultiFrameObjective(pts3, cnt, [v for T in TAGV for v in T], radius)
    f0 = obj(x0)
    x, f, iters, evals = nelder_mead6(obj, x0)
    return dict(T=vec2T(x), x=x, x0=x0, fvals=[f0, f], iters=iters, evals=evals, TAGV=TAGV)
