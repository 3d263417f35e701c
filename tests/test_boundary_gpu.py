"""The drop-in boundary beyond the point tables: the cylinder module's folder entry point, colour input,
rows_updated / cols_updated, the .mat hand-off."""
import json
import os

import numpy as np
import pytest
import torch


def _frames(h, w, n, seed):
    from cpe_amd import synth
    return synth.render_batch(n, h, w, seed=seed, with_gt=False)


@pytest.mark.gpu
def test_cylinder_entry_module_folder(cpe, orc, gpu, tmp_path):
    """python_grid_detection_cylinder.process_images_in_folder (reference :12-64): camera JSON -> undistort -> detect_grid
    -> <stem>_arc.png + processed_images_data.json.  Three stereo pairs, a colour file among them, chunked (chunk 4 < 6
    images), one map per camera; every image equals oracle undistort + oracle detect_grid."""
    import importlib
    import oracle
    from oracle import stages as S
    from PIL import Image
    b = _frames(480, 640, 3, 0)
    K = [[992.0, 0, 322.5], [0, 992.0, 237.5], [0, 0, 1]]
    camL = dict(IntrinsicMatrix=K, RadialDistortion=[0.012, -0.004], TangentialDistortion=[0.0003, -0.0002])
    camR = dict(IntrinsicMatrix=K, RadialDistortion=[-0.008, 0.002], TangentialDistortion=[0.0, 0.0001])
    (tmp_path / 'cam.json').write_text(json.dumps(dict(LeftCamera=camL, RightCamera=camR)))
    src = tmp_path / 'in'; src.mkdir()
    names = ['00', '-1-4', '25']                           # <pan><tilt> stems of getUniqueName.m
    for i, nm in enumerate(names):
        L = b['left'][i].numpy(); R = b['right'][i].numpy()
        if i == 1:                                         # a true colour file: blue-ish tint, channels differ
            rgb = np.stack([(L * 0.9).astype(np.uint8), L, np.minimum(255, L.astype(np.int32) + 3).astype(np.uint8)], 2)
            Image.fromarray(rgb).save(src / f'{nm}L.png')
        else:
            Image.fromarray(L).save(src / f'{nm}L.png')
        Image.fromarray(R).save(src / f'{nm}R.png')
    (src / 'notes.txt').write_text('not an image')
    mod = importlib.import_module('python_grid_detection_cylinder')
    from cpe_amd import folder
    out = tmp_path / 'out'
    res = json.loads(folder.run_folder(str(tmp_path / 'cam.json'), str(src), str(out), target='cylinder', chunk=4))
    assert set(res) == {f'{nm}{c}' for nm in names for c in 'LR'}
    assert json.loads((out / 'processed_images_data.json').read_text()) == res
    for nm in names:
        assert (out / f'{nm}L_arc.png').exists() and (out / f'{nm}R_arc.png').exists()
    # the module function itself (default chunking), same result
    res2 = json.loads(mod.process_images_in_folder(str(tmp_path / 'cam.json'), str(src), str(tmp_path / 'out2')))
    assert res2 == res
    for i, nm in enumerate(names):
        for cam, prm, img in (('L', camL, b['left'][i].numpy()), ('R', camR, b['right'][i].numpy())):
            dist = np.array(prm['RadialDistortion'] + prm['TangentialDistortion'])
            if cam == 'L' and i == 1:
                bgr = np.stack([np.minimum(255, img.astype(np.int32) + 3).astype(np.uint8), img, (img * 0.9).astype(np.uint8)], 2)
                und = np.stack([oracle.undistort(np.ascontiguousarray(bgr[..., c]), np.array(K), dist) for c in range(3)], 2)
                ref = S.detect_grid_bgr(und)                # a colour file takes the colour path (LAB L, per-channel 7x7 blur)
            else:
                ref = S.detect_grid(oracle.undistort(img, np.array(K), dist))
            got = res[f'{nm}{cam}']
            assert ref['status'] == 0
            assert [p['id'] for p in got['points']] == ref['id'].tolist(), (nm, cam)
            assert np.array_equal(np.array([[p['x'], p['y']] for p in got['points']]), ref['xy']), (nm, cam)
            assert got['center_point'] == ref['center'].tolist()
    # a frame the detector fails on stops the run the way the reference's unpack of None does; an empty folder returns None
    Image.fromarray(np.full((480, 640), 7, np.uint8)).save(src / 'zzL.png')
    with pytest.raises(TypeError):
        mod.process_images_in_folder(str(tmp_path / 'cam.json'), str(src), str(tmp_path / 'out3'))
    # ... after the files in front of it have been written, as in the reference's file-by-file loop
    listing = [f for f in os.listdir(src) if f.lower().endswith('.png')]
    before = listing[:listing.index('zzL.png')]
    assert all((tmp_path / 'out3' / (os.path.splitext(f)[0] + '_arc.png')).exists() for f in before)
    assert not (tmp_path / 'out3' / 'processed_images_data.json').exists()
    empty = tmp_path / 'empty'; empty.mkdir()
    assert mod.process_images_in_folder(str(tmp_path / 'cam.json'), str(empty)) is None
    (src / 'zzL.png').unlink()
    Image.fromarray(b['left'][0].numpy()).save(src / 'camera0.png')          # neither L nor R in the name
    with pytest.raises(ValueError):
        mod.process_images_in_folder(str(tmp_path / 'cam.json'), str(src), str(tmp_path / 'out4'))


@pytest.mark.gpu
def test_line_tables_refuse_a_workspace_that_has_moved_on(cpe, orc, gpu):
    """a result keeps a reference to its workspace; once that workspace has served another call (of any size) the line
    tables of the older result are gone, and asking for them is an error instead of garbage"""
    f = _frames(480, 640, 2, 0)
    frames = torch.cat([f['left'], f['right']]).to(gpu)
    ws = cpe.api.DetectWorkspace(4, 480, 640, frames.device)
    det4 = cpe.api.detect_grid_batch(frames, ws)
    rows4, cols4 = cpe.api.line_tables(det4, 3)
    det1 = cpe.api.detect_grid_batch(frames[3:4], ws)             # the ragged last chunk of a folder: same buffer, new layout
    rows1, cols1 = cpe.api.line_tables(det1, 0)
    assert rows1 == rows4 and cols1 == cols4
    with pytest.raises(RuntimeError):
        cpe.api.line_tables(det4, 3)


@pytest.mark.gpu
def test_colour_input_and_error_behaviour(cpe, orc, gpu):
    """BGR2GRAY kernel == oracle on random colour data (incl. a ragged tail); detect_grid on a colour frame == the oracle's
    colour path; the drop-in detect_grid never raises (reference :111-112)."""
    import importlib
    from oracle import stages as S
    rng = np.random.default_rng(5)
    for shape in ((3, 37, 53, 3), (1, 480, 640, 3)):
        bgr = rng.integers(0, 256, shape, dtype=np.uint8)
        got = cpe.api.bgr_to_gray(torch.from_numpy(bgr).to(gpu)).cpu().numpy()
        for k in range(shape[0]):
            assert np.array_equal(got[k], S.bgr2gray(bgr[k]))
    g = rng.integers(0, 256, (1, 16, 16), dtype=np.uint8)
    assert np.array_equal(cpe.api.bgr_to_gray(torch.from_numpy(np.repeat(g[..., None], 3, 3)).to(gpu)).cpu().numpy(), g)
    f = _frames(480, 640, 1, 0)['left'][0].numpy()
    tint = np.stack([np.minimum(255, f.astype(np.int32) + 4).astype(np.uint8), f, (f * 0.93).astype(np.uint8)], 2)
    mod = importlib.import_module('python_grid_detection_cylinder')
    out = mod.detect_grid(tint)
    ref = S.detect_grid_bgr(tint)
    assert out is not None and ref['status'] == 0
    d = json.loads(out[1])
    assert [p['id'] for p in d['points']] == ref['id'].tolist()
    assert np.array_equal(np.array([[p['x'], p['y']] for p in d['points']]), ref['xy'])
    assert mod.detect_grid(np.zeros((480, 640), np.float32)) is None          # wrong dtype: printed, not raised
    assert mod.detect_grid(np.zeros((4, 480, 640, 3), np.uint8)) is None      # wrong rank
    assert mod.detect_grid(np.zeros((16, 16), np.uint8)) is None              # C-ABI argument error (CpeError) is caught too


@pytest.mark.gpu
def test_true_colour_frames(cpe, orc, gpu):
    """a colour camera's view of a red laser grid (R strong, G and B weak, channels with their own noise): the two places
    where the reference looks at the colour planes again -- the L channel of BGR2LAB in detect_largest_blob
    (util_cylinder.py:1840) and the per-channel 7x7 blur of indexing_data (:1433-1435) -- are what the colour entry point
    computes: CLAHE plane, status, centre and every point equal the oracle's colour path; and the colour path is not the
    luma path in disguise (the L plane differs from L of the luma on a large share of the pixels)."""
    from oracle import stages as S
    rng = np.random.default_rng(12)
    f = _frames(480, 640, 2, 3)
    frames = []
    for g in (f['left'][0].numpy(), f['right'][1].numpy()):
        r = g.astype(np.int32)
        bgr = np.stack([r * 0.18 + rng.integers(0, 6, g.shape), r * 0.35 + rng.integers(0, 6, g.shape), r + rng.integers(-2, 3, g.shape)], 2)
        bgr[g >= 245] = g[g >= 245][:, None]                  # the saturated spot blooms white on every channel
        frames.append(np.clip(bgr, 0, 255).astype(np.uint8))
    frames = np.stack(frames)
    det = cpe.api.detect_grid_batch(torch.from_numpy(frames).to(gpu))
    torch.cuda.synchronize()
    clahe = det['ws'].plane('clahe').cpu().numpy()
    for i in range(2):
        ref = S.detect_grid_bgr(frames[i])
        L = S.lab_l_bgr(frames[i])
        assert (L != S.lab_l(S.bgr2gray(frames[i]))).mean() > 0.2
        assert np.array_equal(clahe[i], S.clahe(L))
        assert int(det['status'][i]) == ref['status'] == 0
        m = int(det['n'][i])
        assert m == len(ref['xy']) and m > 20
        assert np.array_equal(det['id'][i, :m].cpu().numpy(), ref['id'])
        assert np.array_equal(det['xy'][i, :m].cpu().numpy(), ref['xy'])
        assert np.array_equal(det['center'][i].cpu().numpy(), ref['center'])
    # grey-replicated colour frames: exactly the grey path
    g = f['left'][1].numpy()
    a = cpe.api.detect_grid_batch(torch.from_numpy(np.repeat(g[None, ..., None], 3, 3)).to(gpu))
    b = cpe.api.detect_grid_batch(torch.from_numpy(g[None]).to(gpu))
    assert int(a['status'][0]) == int(b['status'][0]) == 0 and torch.equal(a['xy'], b['xy']) and torch.equal(a['id'], b['id'])


@pytest.mark.gpu
def test_rows_cols_updated_match_oracle(cpe, orc, gpu):
    """third and fourth return value of detect_grid: every line's intersection list (loop order) and equation"""
    from oracle import stages as S
    b = _frames(480, 640, 2, 4)
    for img in (b['left'][0].numpy(), b['right'][1].numpy()):
        col_img, result_json, rows, cols = cpe.api.detect_grid(img)
        ref = S.detect_grid(img, lines=True)
        assert ref['status'] == 0
        for got, want in ((rows, ref['rows']), (cols, ref['cols'])):
            assert list(got['points'].keys()) == list(want['points'].keys())
            assert list(got['equations'].keys()) == list(want['equations'].keys())
            for k in want['points']:
                assert got['points'][k] == [tuple(p) for p in want['points'][k]], k
                assert got['equations'][k] == list(want['equations'][k]), k
        assert len(rows['points']) >= 4 and len(cols['points']) >= 4
        assert all(len(e) == 6 for e in rows['equations'].values())
        # every JSON point is one of the column points
        pts = {(p['x'], p['y']) for p in json.loads(result_json)['points']}
        assert pts <= {p for v in cols['points'].values() for p in v}


@pytest.mark.gpu
def test_save_mat_round_trip(cpe, orc, gpu, tmp_path):
    """the .mat hand-off: gridPtsPair{.center_point 2x1, .points Nx4} per image (makePyGridPts.m:39-41) and pts3 / cylParams /
    cylT / fvals / meanError per frame (fitSingleCylinder.m:1), readable by scipy.io.loadmat = MATLAB's load"""
    from scipy.io import loadmat
    from cpe_amd import fit
    n = 2
    b = _frames(480, 640, n, 0)
    det = cpe.api.detect_grid_batch(torch.cat([b['left'], b['right']]).to(gpu))
    g1 = fit.GridTables(det['xy'][:n], det['id'][:n], det['n'][:n])
    g2 = fit.GridTables(det['xy'][n:], det['id'][n:], det['n'][n:])
    out = fit.fit_single_cylinder_batch(g1, g2, b['K1'], b['K2'], b['T21'], b['radius'])
    left = [cpe.api.grid_struct(det, i) for i in range(n)]
    right = [cpe.api.grid_struct(det, n + i) for i in range(n)]
    path = cpe.api.save_mat(str(tmp_path / 'frames.mat'), left, right, out, names=['00', '-1-4'])
    m = loadmat(path, squeeze_me=False)
    pair = m['gridPtsPair']
    assert pair.shape == (n, 2) and set(pair.dtype.names) == {'center_point', 'points'}
    fr = m['frames']
    assert fr.shape == (1, n) and set(fr.dtype.names) == {'pts3', 'cylParams', 'cylT', 'fvals', 'meanError', 'status'}
    for i in range(n):
        for c, k in ((0, i), (1, n + i)):
            cnt = int(det['n'][k])
            P = pair[i, c]['points']
            assert P.shape == (cnt, 4) and pair[i, c]['center_point'].shape == (2, 1)
            assert np.array_equal(P[:, :2], det['xy'][k, :cnt].cpu().numpy())
            assert np.array_equal(P[:, 2:], det['id'][k, :cnt].cpu().numpy().astype(np.float64))
            assert np.array_equal(pair[i, c]['center_point'].ravel(), det['center'][k].cpu().numpy())
        mi = int(out['m'][i])
        assert fr[0, i]['pts3'].shape == (3, mi)
        assert np.array_equal(fr[0, i]['pts3'].T, out['pts3'][i, :mi].cpu().numpy())
        assert np.array_equal(fr[0, i]['cylParams'], out['cyl'][i].cpu().numpy())
        assert np.array_equal(fr[0, i]['cylT'], out['T'][i].cpu().numpy().reshape(4, 4))
        assert np.array_equal(fr[0, i]['fvals'].ravel(), out['fvals'][i].cpu().numpy())
        assert float(np.ravel(fr[0, i]['meanError'])[0]) == float(out['mean_err'][i])
    assert [str(x[0][0]) for x in m['names']] == ['00', '-1-4']
