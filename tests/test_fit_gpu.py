"""Geometric half on the GPU (through the C ABI) vs the oracle: same seeded tables, f64 bit-exact
(the kernels and the oracle share the 64-lane reduction tree; stated tolerance = 0)."""
import numpy as np
import pytest
import torch


def make_tables(seed, n, noise=0.05, drop=0.1, h=480, w=640, outlier=0.0):
    """grid-point tables derived from rendered-scene ground truth + pixel noise + missing points"""
    from cpe_amd import synth
    sc = synth.Scene(h=h, w=w)
    K1, K2, T21, Tp = synth.make_rig(sc)
    fp = synth._frame_params(sc, n, seed)
    gt = synth.ground_truth(sc, K1, K2, T21, Tp, fp, sc.pitch_px / sc.focal)
    rng = np.random.default_rng(seed + 99)
    t1, t2 = [], []
    for g in gt:
        m = len(g['idx'])
        k1 = rng.random(m) > drop; k2 = rng.random(m) > drop
        a = np.concatenate([g['uv1'] + noise * rng.standard_normal((m, 2)), g['idx']], 1)[k1]
        b = np.concatenate([g['uv2'] + noise * rng.standard_normal((m, 2)), g['idx']], 1)[k2]
        if outlier:
            bad = rng.random(len(b)) < outlier
            b[bad, :2] += rng.uniform(-3, 3, (bad.sum(), 2))
        t1.append(a[rng.permutation(len(a))]); t2.append(b[rng.permutation(len(b))])
    return t1, t2, K1, K2, T21, fp, sc


@pytest.mark.gpu
@pytest.mark.parametrize('selector,th', [(0, 0.3), (1, 0.5), (2, 0.0), (0, 1e-9)])
def test_select_triangulate_bit_exact(cpe, orc, gpu, selector, th):
    from cpe_amd import fit
    t1, t2, K1, K2, T21, fp, sc = make_tables(3, 6, h=1200, w=1920, outlier=0.1)
    g1 = fit.GridTables.from_lists(t1, gpu); g2 = fit.GridTables.from_lists(t2, gpu)
    out = fit.select_triangulate_batch(g1, g2, K1, K2, T21, selector=selector, th=th)
    torch.cuda.synchronize()
    for i in range(len(t1)):
        if selector == 0:
            c1, c2, idx, fb = orc.choose_idx(t1[i], t2[i], K1, K2, T21, 3, th)
        elif selector == 1:
            r = orc.fit_single_cylinder(t1[i], t2[i], K1, K2, T21, 45.0, selector=1, th=th)
            c1 = None
        else:
            c1, c2, idx = orc.find_correspondences(t1[i], t2[i]); fb = False
        m = int(out['m'][i])
        if c1 is not None:
            assert m == len(c1)
            assert np.array_equal(out['p1'][i, :m].cpu().numpy(), c1)
            assert np.array_equal(out['p2'][i, :m].cpu().numpy(), c2)
            assert np.array_equal(out['idx'][i, :m].cpu().numpy(), idx)
            assert bool(int(out['flags'][i]) & 1) == fb
            X, err = orc.triangulate(c1, c2, K1, K2, T21)
        else:
            X = r['pts3']
            assert m == len(X)
        assert np.array_equal(out['pts3'][i, :m].cpu().numpy(), X), np.abs(out['pts3'][i, :m].cpu().numpy() - X).max()
    assert out['m'].max() > 30


@pytest.mark.gpu
def test_fit_single_cylinder_bit_exact_and_recovers_axis(cpe, orc, gpu):
    from cpe_amd import fit
    t1, t2, K1, K2, T21, fp, sc = make_tables(5, 8, noise=0.03, h=1200, w=1920)
    g1 = fit.GridTables.from_lists(t1, gpu); g2 = fit.GridTables.from_lists(t2, gpu)
    out = fit.fit_single_cylinder_batch(g1, g2, K1, K2, T21, 45.0)
    torch.cuda.synchronize()
    good = 0
    for i in range(len(t1)):
        r = orc.fit_single_cylinder(t1[i], t2[i], K1, K2, T21, 45.0)
        assert int(out['status'][i]) == r['status'] == 0
        m = int(out['m'][i])
        assert m == len(r['pts3'])
        assert np.array_equal(out['cyl'][i].cpu().numpy(), r['cyl'])
        assert np.array_equal(out['T'][i].cpu().numpy(), r['T'])
        assert np.array_equal(out['fvals'][i].cpu().numpy(), r['fvals'])
        assert float(out['mean_err'][i]) == r['mean_err']
        assert out['iters'][i].tolist() == [r['iters'], r['evals']]
        # ground truth (Nelder-Mead may stagnate on some frames exactly as fminsearch would: count successes)
        d = r['cyl'][1, 3:] / np.linalg.norm(r['cyl'][1, 3:])
        gt = fp['dir'][i] * np.sign(fp['dir'][i][1])
        dist = orc.dist_pts3_to_line(fp['org'][i:i + 1], r['cyl'][1, :3], r['cyl'][1, :3] + r['cyl'][1, 3:])
        good += int(np.degrees(np.arccos(np.clip(d @ gt, -1, 1))) < 0.5 and dist[0] < 0.3)
        assert r['fvals'][1] <= r['fvals'][0]
    assert good >= 5, good


@pytest.mark.gpu
def test_fit_edge_cases(cpe, orc, gpu):
    from cpe_amd import fit
    t1, t2, K1, K2, T21, fp, sc = make_tables(7, 3)
    t1[1] = t1[1][:0]                       # empty table
    t2[2] = t2[2][:2]                       # ragged: 2 matches at most
    g1 = fit.GridTables.from_lists(t1, gpu); g2 = fit.GridTables.from_lists(t2, gpu)
    out = fit.fit_single_cylinder_batch(g1, g2, K1, K2, T21, 45.0)
    torch.cuda.synchronize()
    assert int(out['m'][1]) == 0 and int(out['status'][1]) == 5
    assert int(out['status'][2]) == 5
    assert int(out['status'][0]) == 0


@pytest.mark.gpu
def test_lm_mode_bit_exact_and_close_to_nelder_mead(cpe, orc, gpu):
    """the build's Levenberg-Marquardt fast mode (not in the reference): GPU == oracle restatement bit for bit;
    its minimiser agrees with the Nelder-Mead (reference) answer where Nelder-Mead converged:
    gauge-fixed axis direction within 1e-4 rad and f within 1e-6 relative -- the reference's own stopping tolerance
    (TolX = TolFun = 1e-5) bounds how close two converged answers can be expected to be."""
    from cpe_amd import fit
    t1, t2, K1, K2, T21, fp, sc = make_tables(9, 8, noise=0.05, h=1200, w=1920)
    g1 = fit.GridTables.from_lists(t1, gpu); g2 = fit.GridTables.from_lists(t2, gpu)
    sel = fit.select_triangulate_batch(g1, g2, K1, K2, T21, selector=fit.SEL_JOIN)
    nm = fit.fit_cylinder_batch(sel['pts3'], sel['m'], 45.0, mode=fit.FIT_NELDER_MEAD)
    lm = fit.fit_cylinder_batch(sel['pts3'], sel['m'], 45.0, mode=fit.FIT_LM)
    torch.cuda.synchronize()
    close = 0
    for i in range(len(t1)):
        m = int(sel['m'][i]); X = sel['pts3'][i, :m].cpu().numpy()
        ref = orc.fit_cylinder(X, 45.0, mode=1)
        assert np.array_equal(lm['cyl_raw'][i, 1].cpu().numpy(), ref['cyl'])
        assert np.array_equal(lm['fvals'][i].cpu().numpy(), ref['fvals'])
        assert lm['iters'][i].tolist() == [ref['iters'], ref['evals']]
        a = nm['cyl'][i, 1].cpu().numpy(); b = lm['cyl'][i, 1].cpu().numpy()
        da = a[3:] / np.linalg.norm(a[3:]); db = b[3:] / np.linalg.norm(b[3:])
        fa, fb = float(nm['fvals'][i, 1]), float(lm['fvals'][i, 1])
        assert fb <= fa * (1 + 1e-9) + 1e-12                       # LM never ends above Nelder-Mead's value here
        if np.arccos(np.clip(da @ db, -1, 1)) < 1e-4 and abs(fa - fb) <= 1e-6 * max(fa, 1e-12):
            close += 1
        assert int(lm['iters'][i, 0]) < 50
    assert close >= 6, close


@pytest.mark.gpu
def test_choose_idx_and_triangulate_as_separate_entry_points(cpe, orc, gpu):
    """SURVEY 8b lists chooseIdx and triangulate as entry points of their own: cpe_choose_idx_batch selects (with the fallback
    join), cpe_triangulate_batch takes pairs that are already matched -- what fitSingleCylinder.m:12 and :15-17 do in turn.
    Both equal the oracle and the merged cpe_select_triangulate_batch bit for bit."""
    from cpe_amd import fit
    t1, t2, K1, K2, T21, fp, sc = make_tables(7, 5, h=1200, w=1920, outlier=0.05)
    t1.append(t1[0][:0]); t2.append(t2[0])                       # an empty left table: m = 0
    g1 = fit.GridTables.from_lists(t1, gpu); g2 = fit.GridTables.from_lists(t2, gpu)
    sel = fit.choose_idx_batch(g1, g2, K1, K2, T21, 3, 0.3)
    tri = fit.triangulate_batch(sel['p1'], sel['p2'], sel['m'], K1, K2, T21)
    both = fit.select_triangulate_batch(g1, g2, K1, K2, T21, selector=0, th=0.3)
    torch.cuda.synchronize()
    for i in range(len(t1)):
        m = int(sel['m'][i])
        assert m == int(both['m'][i])
        if len(t1[i]) == 0:
            assert m == 0 and float(tri['mean_err'][i]) == 0.0
            continue
        c1, c2, idx, fb = orc.choose_idx(t1[i], t2[i], K1, K2, T21, 3, 0.3)
        assert m == len(c1) and np.array_equal(sel['p1'][i, :m].cpu().numpy(), c1) and np.array_equal(sel['p2'][i, :m].cpu().numpy(), c2)
        assert np.array_equal(sel['idx'][i, :m].cpu().numpy(), idx) and bool(int(sel['flags'][i]) & 1) == fb
        X, err = orc.triangulate(c1, c2, K1, K2, T21)
        assert np.array_equal(tri['pts3'][i, :m].cpu().numpy(), X)
        assert np.array_equal(tri['err'][i, :m].cpu().numpy(), err)
        assert np.array_equal(tri['pts3'][i, :m].cpu().numpy(), both['pts3'][i, :m].cpu().numpy())
        assert float(tri['mean_err'][i]) == float(both['mean_err'][i])
    assert int(sel['m'].max()) > 100
