"""Row f-1: fitCylinderWPts3sAngs -- host Nelder-Mead + GPU objective vs the oracle's C restatement."""
import math

import numpy as np
import pytest
import torch


def make_scene(F=10, seed=0, noise=0.05, npts=160):
    """cylinders posed by an AGV pan/tilt head: T_C1_cyl = T_true * getTAGVcyl(pan, tilt); points on the camera-facing
    side of each cylinder (3-D, camera-1 frame), as fitSingleCylinder would triangulate them"""
    import oracle
    rng = np.random.default_rng(seed)
    angles = np.stack([rng.uniform(-0.35, 0.35, F), rng.uniform(-0.2, 0.2, F)], 1)
    A0 = oracle.get_TAGVcyl(0.0, 0.0).reshape(4, 4)
    # T_true: AGV frame -> camera frame, chosen so that the cylinder sits ~420 mm in front of the camera, axis ~ +y
    Rz = np.array([[0, -1, 0], [-1, 0, 0], [0, 0, -1.0]]).T            # inverse of T_T2_CYL's rotation
    ang = 0.15
    Rx = np.array([[1, 0, 0], [0, math.cos(ang), -math.sin(ang)], [0, math.sin(ang), math.cos(ang)]])
    Rt = Rx @ Rz
    t = np.array([5.0, -10.0, 420.0]) - Rt @ A0[:3, 3]
    Ttrue = np.eye(4); Ttrue[:3, :3] = Rt; Ttrue[:3, 3] = t
    from cpe_amd.fit import MAXP
    P = np.zeros((F, MAXP, 3)); cnt = np.zeros(F, np.int32)
    for i in range(F):
        Tc = Ttrue @ oracle.get_TAGVcyl(*angles[i]).reshape(4, 4)
        o, a = Tc[:3, 3], Tc[:3, 1]
        toc = -o - (-o @ a) * a; toc /= np.linalg.norm(toc)            # radial direction facing the camera
        b = np.cross(a, toc)
        s = rng.uniform(-55, 55, npts); phi = rng.uniform(-1.0, 1.0, npts)
        pts = o + s[:, None] * a + 45.0 * (np.cos(phi)[:, None] * toc + np.sin(phi)[:, None] * b)
        pts += noise * rng.standard_normal(pts.shape)
        P[i, :npts] = pts; cnt[i] = npts
    return P, cnt, angles, Ttrue


@pytest.mark.gpu
def test_multi_frame_fit_matches_oracle(cpe, orc, gpu):
    from cpe_amd import fit, multiframe
    P, cnt, angles, Ttrue = make_scene()
    Pd = torch.from_numpy(P).to(gpu); cd = torch.from_numpy(cnt).to(gpu)
    per = fit.fit_cylinder_batch(Pd, cd, 45.0)                          # fitCylinderWPts3 per frame (:35)
    torch.cuda.synchronize()
    raw = per['cyl_raw'].cpu().numpy()
    for i in range(len(cnt)):                                           # same per-frame fits as the oracle
        r = orc.fit_cylinder(P[i, :cnt[i]], 45.0)
        assert np.array_equal(raw[i, 0], r['cyl0']) and np.array_equal(raw[i, 1], r['cyl'])
    TAGV = np.stack([orc.get_TAGVcyl(*a) for a in angles])
    ref = orc.multi_fit(P, cnt, TAGV, raw, 45.0)
    got = multiframe.fit_multi_frame(Pd, cd, per['cyl_raw'], angles, 45.0)
    assert np.array_equal(np.array(got['x0']), ref['x0'])
    # objective: GPU terms == oracle, bit for bit, at the initial pose and at a perturbed one
    obj = multiframe.MultiFrameObjective(Pd, cd, TAGV.ravel().tolist(), 45.0)
    for x in (ref['x0'], ref['x0'] + 0.01):
        assert obj(list(x)) == orc.multi_objective(x, P, cnt, TAGV, 45.0)
    assert got['fvals'] == ref['fvals'].tolist()
    assert np.array_equal(np.array(got['x']), ref['x'])
    assert (got['iters'], got['evals']) == (ref['iters'], ref['evals'])
    assert np.array_equal(np.array(got['T']), ref['T'])
    assert got['fvals'][1] <= got['fvals'][0]
