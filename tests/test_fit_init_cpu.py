"""The oracle's initial cylinder (fitCylinderWPts3.m:7-31) against an independent numpy restatement that uses LAPACK for
every solve, as MATLAB does: pca -> SVD of the centred points, fitplane -> eigh of the covariance, fitquadsurf's `A \\ b`
-> lstsq (SVD), eig of the 2x2 -> eigh.  The oracle (and the kernel, which is bit-equal to it) solves the 20 x 5
least-squares problem through its normal equations and takes the 3x3 eigenvectors by Jacobi rotations (DESIGN.md §2,
deviation 8): this test bounds what that costs on the initial guess -- far below the 1e-6 pose tolerance."""
import numpy as np
import pytest

import oracle


def _points(seed, n=240, R=45.0, noise=0.02):
    rng = np.random.default_rng(seed)
    th = rng.uniform(-0.9, 0.9, n); t = rng.uniform(-60, 60, n)
    P = np.stack([R * np.sin(th), t, 500.0 - R * np.cos(th)], 1)          # surface facing the camera, axis ~ y
    a, b = rng.uniform(-0.4, 0.4, 2)
    Rx = np.array([[1, 0, 0], [0, np.cos(a), -np.sin(a)], [0, np.sin(a), np.cos(a)]])
    Rz = np.array([[np.cos(b), -np.sin(b), 0], [np.sin(b), np.cos(b), 0], [0, 0, 1]])
    c = P.mean(0)
    return (P - c) @ (Rz @ Rx).T + c + rng.normal(0, noise, (n, 3))


def _dist_to_line(P, p1, p2):                                               # getDistPts3ToLine.m
    d = (p2 - p1) / np.linalg.norm(p2 - p1)
    v = P - p1
    return np.linalg.norm(v - np.outer(v @ d, d), axis=1)


def _init_numpy(P, R):
    """fitCylinderWPts3.m:7-31 with numpy / LAPACK"""
    ctr = P.mean(0)
    _, _, Vt = np.linalg.svd(P - ctr, full_matrices=False)                 # pca(Pts3'): coeff(:, 3)
    rdir = Vt[2] if Vt[2, 2] >= 0 else -Vt[2]
    i = int(np.argmin(_dist_to_line(P, ctr, ctr + rdir)))
    d2s = np.linalg.norm(ctr - P[i])
    d2 = ((P - P[i]) ** 2).sum(1)
    nb = np.argsort(d2, kind='stable')[:20]                                # knnsearch(..., 'K', 20)
    Q = P[nb]
    w, V = np.linalg.eigh(np.cov(Q.T))                                     # fitplane.m:13-14
    z = V[:, 0]
    if z[2] < 0:
        z = -z                                                             # deviation 2 (sign of the normal)
    x = np.array([1.0, 0, 0])
    if abs(z @ x) > 0.9:
        x = np.array([0, 1.0, 0])
    y = np.cross(z, x); x = np.cross(y, z)
    L = (Q - Q.mean(0)) @ np.stack([x, y, z], 1)
    A = np.stack([L[:, 0] ** 2, L[:, 0] * L[:, 1], L[:, 1] ** 2, L[:, 0], L[:, 1]], 1)
    co = np.linalg.lstsq(A, L[:, 2], rcond=None)[0]                        # A \ b
    _, V2 = np.linalg.eigh(np.array([[2 * co[0], co[1]], [co[1], 2 * co[2]]]))
    kdir = np.stack([x, y], 1) @ V2[:, 0]                                  # K(:, 1, i)
    return ctr + rdir * (R - d2s), kdir, np.linalg.cond(A)


@pytest.mark.parametrize('seed', range(12))
def test_initial_cylinder_matches_lapack_restatement(seed):
    P = _points(seed)
    got = oracle.fit_cylinder(P, 45.0)
    assert got['status'] == 0
    org, kdir, cond = _init_numpy(P, 45.0)
    o0, d0 = got['cyl0'][:3], got['cyl0'][3:]
    if d0 @ kdir < 0:
        kdir = -kdir                                                       # eig's sign is LAPACK's choice; the line is the same
    assert np.abs(o0 - org).max() < 1e-11 * max(1.0, np.abs(org).max())
    assert np.abs(d0 - kdir).max() < 1e-12, (np.abs(d0 - kdir).max(), cond)   # seen: 7e-16 at cond(A) <= 28


@pytest.mark.parametrize('seed', range(8))
def test_nelder_mead_equals_scipy(seed):
    """The oracle's restatement of fminsearch.m (called at fitCylinderWPts3.m:38 with TolX = TolFun = 1e-5) against scipy's
    Nelder-Mead -- an independent implementation of the same published algorithm (Lagarias et al. 1998: 5 % initial simplex,
    rho 1, chi 2, psi 0.5, sigma 0.5, both tolerances required).  Same objective (the oracle's), same start: the two
    walks are identical step by step, so the minimiser, the value and both counters are EQUAL, not close."""
    from scipy.optimize import minimize
    P = _points(100 + seed, noise=0.05)
    got = oracle.fit_cylinder(P, 45.0)
    f = lambda x: oracle.cyl_objective(x, P, 45.0)
    assert f(got['cyl0']) == got['fvals'][0]
    r = minimize(f, got['cyl0'], method='Nelder-Mead', options=dict(xatol=1e-5, fatol=1e-5, maxiter=100000, maxfev=100000))
    assert np.array_equal(r.x, got['cyl'])
    assert r.fun == got['fvals'][1]
    assert (r.nit, r.nfev) == (got['iters'], got['evals'])


def test_triangulation_matches_lapack_svd():
    """MATLAB's `triangulate` (fitSingleCylinder.m:15) is the homogeneous DLT: per point the right singular vector of the
    4 x 4 matrix [x P(3,:) - P(1,:); y P(3,:) - P(2,:)] (both cameras) for the smallest singular value, and the mean of
    the two reprojection distances.  The oracle takes that vector by one-sided Jacobi rotations; here LAPACK's SVD on the
    same matrix, and the exact 3-D points the pixels were projected from."""
    rng = np.random.default_rng(3)
    K1 = np.array([[2400.0, 0, 960], [0, 2400, 600], [0, 0, 1]]); K2 = np.array([[2390.0, 0, 955], [0, 2395, 610], [0, 0, 1]])
    a = 0.12
    T21 = np.eye(4); T21[:3, :3] = [[np.cos(a), 0, -np.sin(a)], [0, 1, 0], [np.sin(a), 0, np.cos(a)]]; T21[:3, 3] = [-120.0, 1.5, 8.0]
    X = np.stack([rng.uniform(-80, 80, 300), rng.uniform(-60, 60, 300), rng.uniform(420, 560, 300)], 1)
    P1 = K1 @ np.eye(4)[:3]; P2 = K2 @ T21[:3]
    h = np.c_[X, np.ones(len(X))]
    p1 = (h @ P1.T); p1 = p1[:, :2] / p1[:, 2:]
    p2 = (h @ P2.T); p2 = p2[:, :2] / p2[:, 2:]
    Xo, err = oracle.triangulate(p1, p2, K1, K2, T21)
    assert np.abs(Xo - X).max() < 1e-8 and err.max() < 1e-9                 # exact pixels: the rendered points come back
    p1n = p1 + rng.normal(0, 0.3, p1.shape); p2n = p2 + rng.normal(0, 0.3, p2.shape)
    Xo, err = oracle.triangulate(p1n, p2n, K1, K2, T21)
    for i in range(len(X)):
        A = np.stack([p1n[i, 0] * P1[2] - P1[0], p1n[i, 1] * P1[2] - P1[1], p2n[i, 0] * P2[2] - P2[0], p2n[i, 1] * P2[2] - P2[1]])
        v = np.linalg.svd(A)[2][-1]
        Xi = v[:3] / v[3]
        assert np.abs(Xo[i] - Xi).max() < 1e-7, (i, Xo[i], Xi)
        q1 = P1 @ np.r_[Xi, 1]; q2 = P2 @ np.r_[Xi, 1]
        e = (np.linalg.norm(p1n[i] - q1[:2] / q1[2]) + np.linalg.norm(p2n[i] - q2[:2] / q2[2])) / 2
        assert abs(err[i] - e) < 1e-8
