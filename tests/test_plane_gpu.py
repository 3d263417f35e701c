"""Row f-2: the planar-target script (python_grid_detection_plane.py over utils/util_plane.py) on the GPU vs its oracle
restatement, stage by stage and end to end (bit-exact, as for the cylinder script).  The oracle's cv2-free line logic is
pinned against the real util_plane functions in tests/test_oracle_golden.py::test_plane_lines_vs_reference."""
import numpy as np
import pytest
import torch


def _plane_frames(h, w, n, seed):
    from cpe_amd import synth
    # a plane = a cylinder of 5 m radius seen from 34 cm: the grid lines are straight to a fraction of a pixel
    sc = synth.Scene(h=h, w=w, radius=5000.0, depth=(5340.0, 5400.0), tilt_deg=4.0)
    b = synth.render_batch(n, h, w, seed=seed, scene=sc, with_gt=False)
    return torch.cat([b['left'], b['right']])


@pytest.mark.gpu
@pytest.mark.parametrize('h,w,seed', [(600, 800, 3), (1200, 1920, 5), (483, 650, 9)])
def test_plane_detect_matches_oracle(cpe, orc, gpu, h, w, seed):
    from oracle import stages as S
    frames = _plane_frames(h, w, 1, seed)
    det = cpe.api.detect_grid_batch(frames.to(gpu), target='plane')
    torch.cuda.synchronize()
    ws = det['ws']
    planes = {k: ws.plane(k).cpu().numpy() for k in ('binary', 'hmask', 'vmask', 'mask_contour', 'roi_h', 'roi_v', 'exp_h', 'exp_v')}
    state = ws.state()
    npy = frames.numpy()
    n_ok = 0
    for i in range(npy.shape[0]):
        ref = S.detect_grid_plane(npy[i], debug=True)
        tag = f'frame {i}'
        for k in ('binary', 'hmask', 'vmask', 'mask_contour'):
            assert np.array_equal(planes[k][i], ref[k]), (tag, k, int((planes[k][i] != ref[k]).sum()))
        assert int(det['status'][i]) == ref['status'], (tag, state[i]['overflow'], ref['status'])
        if ref['status'] == 1:
            continue
        assert (state[i]['rect0'], state[i]['rect1'], state[i]['rect2'], state[i]['rect3']) == tuple(ref['rect']), tag
        if ref['status'] == 2:
            continue
        assert state[i]['r0'] == ref['r0'] and (state[i]['spot0'], state[i]['spot1'], state[i]['spot2'], state[i]['spot3']) == tuple(ref['spot']), tag
        assert state[i]['n_joints'] == ref['n_cyl_joints'], tag
        for k in ('roi_h', 'roi_v', 'exp_h', 'exp_v'):
            assert np.array_equal(planes[k][i], ref[k]), (tag, k, int((planes[k][i] != ref[k]).sum()))
        if ref['status'] != 0:
            continue
        n_ok += 1
        assert (state[i]['n_rows'], state[i]['n_cols']) == (ref['n_rows'], ref['n_cols']), tag
        m = int(det['n'][i])
        assert m == len(ref['xy']), tag
        assert np.array_equal(det['id'][i, :m].cpu().numpy(), ref['id']), tag
        assert np.array_equal(det['center'][i].cpu().numpy(), ref['center']), tag
        got = det['xy'][i, :m].cpu().numpy()
        assert np.array_equal(got, ref['xy']), (tag, np.abs(got - ref['xy']).max())
        assert m >= 60 and ref['id'][:, 1].min() < 0      # a real grid, and the negative columns are kept (no remove_minus_labels)
    assert n_ok >= 1


@pytest.mark.gpu
def test_plane_failure_and_argument_checks(cpe, orc, gpu):
    from oracle import stages as S
    dark = torch.full((1, 480, 640), 20, dtype=torch.uint8)
    det = cpe.api.detect_grid_batch(dark.to(gpu), target='plane')
    assert int(det['status'][0]) == 1 == S.detect_grid_plane(dark[0].numpy())['status']       # get_convex_hull raises: no contour
    with pytest.raises(Exception):
        cpe.api.detect_grid_batch(dark.to(gpu), target='plane', subpixel=True)


@pytest.mark.gpu
def test_plane_broken_columns_are_merged(cpe, orc, gpu):
    """dark patches over a few vertical lines, longer than the 201-px expansion can bridge: those columns fall into two
    components each, the fit step merges the two short pieces (util_plane.py:452-523).  GPU == oracle, and the merge
    really happened (more components than columns)."""
    from oracle import stages as S
    frames = _plane_frames(1200, 1920, 1, 5)[:1].clone()
    ref0 = S.detect_grid_plane(frames[0].numpy(), debug=True)
    assert ref0['status'] == 0
    x0, y0, rw, rh = ref0['rect']
    g = frames[0].numpy().copy()
    cx = x0 + rw // 2
    for dx in (-260, -130, 170):                      # three patches, each 330 px tall and 30 px wide, over the grid's middle rows
        xs = cx + dx
        g[y0 + rh // 2 - 165:y0 + rh // 2 + 165, xs:xs + 30] = np.minimum(g[y0 + rh // 2 - 165:y0 + rh // 2 + 165, xs:xs + 30], 14)
    frames[0] = torch.from_numpy(g)
    ref = S.detect_grid_plane(g, debug=True)
    det = cpe.api.detect_grid_batch(frames.to(gpu), target='plane')
    torch.cuda.synchronize()
    st = det['ws'].state()[0]
    assert int(det['status'][0]) == ref['status'] == 0
    for k in ('mask_contour', 'exp_h', 'exp_v'):
        assert np.array_equal(det['ws'].plane(k)[0].cpu().numpy(), ref[k]), k
    m = int(det['n'][0])
    assert m == len(ref['xy']) and np.array_equal(det['xy'][0, :m].cpu().numpy(), ref['xy'])
    assert np.array_equal(det['id'][0, :m].cpu().numpy(), ref['id'])
    assert (st['n_rows'], st['n_cols']) == (ref['n_rows'], ref['n_cols'])
    # components of the vertical mask that hold joints vs columns after the fit
    x, y, w_, h_ = ref['rect']
    lab = S.connected_components(ref['exp_v'][y:y + h_, x:x + w_])
    labs = lab[1] if isinstance(lab, tuple) else lab
    joints = [(jx, jy) for jx, jy in ref['joints'] if x <= jx < x + w_ and y <= jy < y + h_]
    used = {int(labs[jy - y, jx - x]) for jx, jy in joints if labs[jy - y, jx - x] > 0}
    assert len(used) > ref['n_cols'], (len(used), ref['n_cols'])


@pytest.mark.gpu
def test_plane_entry_module_folder(cpe, orc, gpu, tmp_path):
    """python_grid_detection_plane.process_images_in_folder: camera JSON -> undistort (row f-3) -> planar detect_grid ->
    <stem>_arc.png + processed_images_data.json, as the reference's CLI does (python_grid_detection_plane.py:13-70)"""
    import json, importlib
    from PIL import Image
    from oracle import stages as S
    import oracle
    frames = _plane_frames(600, 800, 1, 3).numpy()
    K = [[1240.0, 0, 400.0], [0, 1240.0, 300.0], [0, 0, 1]]
    cam = dict(IntrinsicMatrix=K, RadialDistortion=[0.01, -0.003], TangentialDistortion=[0.0002, -0.0001])
    (tmp_path / 'cam.json').write_text(json.dumps(dict(LeftCamera=cam, RightCamera=cam)))
    src = tmp_path / 'in'; src.mkdir()
    Image.fromarray(frames[0]).save(src / 'img_L_000.png')
    Image.fromarray(frames[1]).save(src / 'img_R_000.png')
    mod = importlib.import_module('python_grid_detection_plane')
    out = tmp_path / 'out'
    res = json.loads(mod.process_images_in_folder(str(tmp_path / 'cam.json'), str(src), str(out)))
    assert set(res) == {'img_L_000', 'img_R_000'} and (out / 'img_L_000_arc.png').exists()
    saved = json.loads((out / 'processed_images_data.json').read_text())
    assert saved == res
    # the same through the oracle: undistort, then the planar detector
    und = oracle.undistort(frames[0], np.array(K), np.array([0.01, -0.003, 0.0002, -0.0001]))
    ref = S.detect_grid_plane(und)
    pts = res['img_L_000']['points']
    assert ref['status'] == 0 and len(pts) == len(ref['xy'])
    assert [p['id'] for p in pts] == ref['id'].tolist()
    assert np.array_equal(np.array([[p['x'], p['y']] for p in pts]), ref['xy'])
    assert res['img_L_000']['center_point'] == ref['center'].tolist()


@pytest.mark.gpu
def test_more_than_64_lines_in_one_direction(cpe, orc, gpu):
    """85 columns cross every row: per-line joint and intersection lists longer than 64 entries (the first sizing of the
    tables truncated them silently).  Both scripts, GPU == oracle; an overflow would have to say so (status 6)."""
    from cpe_amd import synth
    from oracle import stages as S
    h, w = 520, 2700
    sc = synth.Scene(h=h, w=w, radius=5000.0, depth=(5340.0, 5400.0), tilt_deg=2.0, pitch_px=30.0, half_lines=42, half_lines_v=4)
    b = synth.render_batch(1, h, w, seed=4, scene=sc, with_gt=False)
    frames = torch.cat([b['left'], b['right']])
    for target, fn in (('plane', S.detect_grid_plane), ('cylinder', S.detect_grid)):
        det = cpe.api.detect_grid_batch(frames.to(gpu), target=target)
        torch.cuda.synchronize()
        state = det['ws'].state()
        for i in range(2):
            ref = fn(frames[i].numpy(), debug=True)
            assert ref['status'] == 0 and ref['n_cols'] > 64
            assert int(det['status'][i]) == 0, (target, i, state[i]['overflow'])
            assert (state[i]['n_rows'], state[i]['n_cols']) == (ref['n_rows'], ref['n_cols'])
            m = int(det['n'][i])
            assert m == len(ref['xy'])
            assert np.array_equal(det['id'][i, :m].cpu().numpy(), ref['id'])
            assert np.array_equal(det['xy'][i, :m].cpu().numpy(), ref['xy'])
            assert np.array_equal(det['center'][i].cpu().numpy(), ref['center'])
