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
