"""BUILD-DEFINED extension (BASELINE config 5): RANSAC around fitCylinderWPts3.  No reference counterpart; the GPU kernel is
compared with its C restatement bit for bit, and checked for what it is for: outliers in the 3-D points."""
import numpy as np
import pytest
import torch


def _cyl_points(rng, n, R=45.0, noise=0.03, n_out=0):
    """points on a cylinder of radius R (axis roughly along y, in front of the camera) + gross outliers"""
    th = rng.uniform(-0.9, 0.9, n) + np.pi          # camera sees the near side
    yy = rng.uniform(-60, 60, n)
    P = np.stack([R * np.sin(th), yy, 450.0 + R * np.cos(th)], 1)
    a = np.deg2rad(rng.uniform(-8, 8)); b = np.deg2rad(rng.uniform(-8, 8))
    Rx = np.array([[1, 0, 0], [0, np.cos(a), -np.sin(a)], [0, np.sin(a), np.cos(a)]])
    Rz = np.array([[np.cos(b), -np.sin(b), 0], [np.sin(b), np.cos(b), 0], [0, 0, 1]])
    c = np.array([0, 0, 450.0])
    P = (P - c) @ (Rz @ Rx).T + c + rng.normal(0, noise, (n, 3))
    axis = (Rz @ Rx) @ np.array([0, 1.0, 0])
    if n_out:
        idx = rng.choice(n, n_out, replace=False)
        P[idx] += rng.normal(0, 12.0, (n_out, 3))
    return P, axis


@pytest.mark.gpu
@pytest.mark.parametrize('mode', [1, 0])
def test_ransac_matches_oracle_and_rejects_outliers(cpe, orc, gpu, mode):
    rng = np.random.default_rng(5)
    cases = [(260, 0), (240, 30), (200, 50), (90, 10), (5, 0), (2, 0)]
    n = len(cases)
    X = np.zeros((n, cpe.fit.MAXP, 3)); cnt = np.zeros(n, np.int32); axes = []
    for i, (m, no) in enumerate(cases):
        P, ax = _cyl_points(rng, max(m, 1), n_out=no)
        X[i, :m] = P[:m]; cnt[i] = m; axes.append(ax)
    kw = dict(hypotheses=48, sample=10, tau=0.4, seed=1234, hyp_iters=8)
    maxit = 100000 if mode == 1 else 400
    out = cpe.fit.fit_cylinder_ransac_batch(torch.from_numpy(X).to(gpu), torch.from_numpy(cnt).to(gpu), 45.0, frame0=7,
                                            mode=mode, max_iter=maxit, max_fun_evals=maxit, **kw)
    plain = cpe.fit.fit_cylinder_batch(torch.from_numpy(X).to(gpu), torch.from_numpy(cnt).to(gpu), 45.0, mode=1)
    torch.cuda.synchronize()
    for i, (m, no) in enumerate(cases):
        if m < 3:
            assert int(out['status'][i]) == 5 and int(out['n_inliers'][i]) == 0
            continue
        ref = orc.fit_cylinder_ransac(X[i, :m], 45.0, frame=7 + i, mode=mode, maxiter=maxit, maxfun=maxit, **kw)
        assert ref['status'] == 0 and int(out['status'][i]) == 0
        assert int(out['n_inliers'][i]) == ref['n_inliers'], i
        mask = out['inlier_mask'][i, :m].cpu().numpy()
        assert np.array_equal(mask, ref['mask']), i
        assert np.array_equal(out['cyl_raw'][i, 0].cpu().numpy(), ref['cyl0']), i
        assert np.array_equal(out['cyl_raw'][i, 1].cpu().numpy(), ref['cyl']), i
        assert np.array_equal(out['fvals'][i].cpu().numpy(), ref['fvals']), i
        assert out['iters'][i].tolist() == [ref['iters'], ref['evals']], i
        Q = X[i, :m][ref['mask'] > 0]
        assert np.array_equal(out['cyl'][i, 1].cpu().numpy(), orc.apply_prior(ref['cyl'], Q)), i
        if m >= 90:
            # what it is for: the consensus set holds the clean points, and the axis beats the plain fit when there are outliers
            assert int(out['n_inliers'][i]) >= 0.9 * (m - no), (i, int(out['n_inliers'][i]))
            d = out['cyl_raw'][i, 1, 3:].cpu().numpy(); d = d / np.linalg.norm(d)
            ang = np.degrees(np.arccos(min(1.0, abs(float(d @ axes[i])))))
            dp = plain['cyl_raw'][i, 1, 3:].cpu().numpy(); dp = dp / np.linalg.norm(dp)
            angp = np.degrees(np.arccos(min(1.0, abs(float(dp @ axes[i])))))
            assert ang < 0.5, (i, ang)
            if no:
                assert ang <= angp + 1e-9, (i, ang, angp)
