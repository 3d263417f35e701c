"""Image half on the GPU (through cpe_detect_grid_batch) vs the oracle, stage by stage and end to end.
Integer / byte stages: bit-exact.  Point coordinates: the kernels run the oracle's f64 operation order,
so they are compared for equality as well (stated tolerance 0; BASELINE allows 1e-3 px)."""
import os

import numpy as np
import pytest
import torch


def _frames(h, w, n, seed):
    from cpe_amd import synth
    b = synth.render_batch(n, h, w, seed=seed, with_gt=False)
    return torch.cat([b['left'], b['right']])


def _compare(cpe, orc, gpu, frames, check_planes=True, allow_overflow=None):
    """allow_overflow: {frame index: bit of FrameState.overflow} -- the ONLY frames that may end with CPE_ST_OVERFLOW where
    the oracle has another status, each for the named capacity of the library's own workspace (cpe_dev.h OVF_*)"""
    from oracle import stages as S
    det = cpe.api.detect_grid_batch(frames.to(gpu))
    torch.cuda.synchronize()
    ws = det['ws']
    planes = {k: ws.plane(k).cpu().numpy() for k in ('binary', 'hmask', 'vmask', 'mask_contour', 'roi_h', 'roi_v',
                                                     'exp_h', 'exp_v', 'clahe')} if check_planes else {}
    joints = ws.plane('joints').cpu().numpy()
    state = ws.state()
    npy = frames.numpy()
    n_ok = 0
    for i in range(npy.shape[0]):
        ref = S.detect_grid(npy[i], debug=True)
        tag = f'frame {i}'
        if check_planes:
            assert np.array_equal(planes['binary'][i], ref['binary']), tag
            assert np.array_equal(planes['hmask'][i], ref['hmask']), tag
            assert np.array_equal(planes['vmask'][i], ref['vmask']), tag
            assert np.array_equal(planes['clahe'][i], S.clahe(S.lab_l(npy[i]))), tag
            assert np.array_equal(planes['mask_contour'][i], ref['mask_contour']), tag
        if allow_overflow and i in allow_overflow and int(det['status'][i]) == 6:
            assert state[i]['overflow'] == allow_overflow[i], (tag, state[i]['overflow'])
            continue            # build-defined: that capacity of the workspace was exceeded (pathological frame)
        assert int(det['status'][i]) == ref['status'], (tag, state[i], ref['status'])
        if ref['status'] in (1,):
            continue
        assert (state[i]['rect0'], state[i]['rect1'], state[i]['rect2'], state[i]['rect3']) == tuple(ref['rect']), tag
        assert state[i]['n_kp'] == ref['n_keypoints'], tag
        if ref['status'] == 2:
            continue
        assert state[i]['r0'] == ref['r0'] and (state[i]['spot0'], state[i]['spot1'], state[i]['spot2'], state[i]['spot3']) == tuple(ref['spot']), tag
        assert state[i]['n_joints'] == ref['n_cyl_joints'], tag
        if check_planes:
            for k in ('roi_h', 'roi_v', 'exp_h', 'exp_v'):
                assert np.array_equal(planes[k][i], ref[k]), (tag, k, int((planes[k][i] != ref[k]).sum()))
        if ref['status'] != 0:
            continue
        n_ok += 1
        m = int(det['n'][i])
        assert m == len(ref['xy']), tag
        assert np.array_equal(det['id'][i, :m].cpu().numpy(), ref['id']), tag
        assert np.array_equal(det['center'][i].cpu().numpy(), ref['center']), tag
        got = det['xy'][i, :m].cpu().numpy()
        assert np.array_equal(got, ref['xy']), (tag, np.abs(got - ref['xy']).max())
    return n_ok


@pytest.mark.gpu
@pytest.mark.parametrize('h,w,seed', [(480, 640, 0), (600, 800, 4), (602, 801, 6), (483, 650, 8)])
def test_detect_small_frames_stage_by_stage(cpe, orc, gpu, h, w, seed):
    # 602x801 / 483x650: sizes that are not multiples of 4 (CLAHE pads with BORDER_REFLECT_101) nor of any tile size
    n_ok = _compare(cpe, orc, gpu, _frames(h, w, 2, seed))
    assert n_ok >= 2


@pytest.mark.gpu
@pytest.mark.parametrize('seed', [20, 21, 22, 23, 24, 25])
def test_detect_seed_sweep_with_degraded_frames(cpe, orc, gpu, seed):
    """other scenes, dimmer exposure, extra sensor noise, a dead band: whatever the reference would make of the frame
    (points or one of its failure statuses), the GPU path makes the same of it"""
    h, w = ((480, 640), (600, 800), (512, 768))[seed % 3]
    f = _frames(h, w, 1, seed).numpy().astype(np.int32)
    rng = np.random.default_rng(seed)
    f[0] = f[0] * (0.7 + 0.05 * (seed % 5))                                  # dimmer exposure
    f[1] = f[1] + rng.integers(-6, 7, size=f[1].shape)                        # extra noise
    if seed % 2:
        f[1][:, w // 2 - 3:w // 2 + 3] = 0                                    # a dead band through the grid
    frames = torch.from_numpy(np.clip(f, 0, 255).astype(np.uint8))
    _compare(cpe, orc, gpu, frames, check_planes=True)


@pytest.mark.gpu
def test_detect_full_size(cpe, orc, gpu):
    from cpe_amd import synth
    b = synth.render_batch(2, 1200, 1920, seed=11, device='cuda', with_gt=False)
    frames = torch.cat([b['left'], b['right']]).cpu()
    n_ok = _compare(cpe, orc, gpu, frames)
    assert n_ok == 4


@pytest.mark.gpu
def test_detect_4k_frame(cpe, orc, gpu):
    """3840x2160 (BASELINE config 5 frame size): wider than the 2048-column LDS path of the hull kernel"""
    from cpe_amd import synth
    b = synth.render_batch(1, 2160, 3840, seed=17, device='cuda', with_gt=False)
    frames = torch.cat([b['left'], b['right']]).cpu()
    n_ok = _compare(cpe, orc, gpu, frames)
    assert n_ok >= 1


@pytest.mark.gpu
def test_detect_failure_statuses(cpe, orc, gpu):
    """frames on which the reference raises inside detect_grid: all-dark (no blob region), grid without the
    saturated spot, pure noise -- the batch is never aborted, every frame gets the oracle's status."""
    rng = np.random.default_rng(3)
    f = _frames(480, 640, 1, 2).numpy().copy()
    dark = np.full((480, 640), 7, np.uint8)
    nospot = f[0].copy(); nospot[nospot > 235] = 200
    noise = rng.integers(0, 60, size=(480, 640), dtype=np.uint8)       # tens of thousands of specks: may overflow
    faint = rng.integers(5, 12, size=(480, 640), dtype=np.uint8)
    frames = torch.from_numpy(np.stack([dark, nospot, noise, f[1], faint]))
    # frame 2 (uniform noise 0..59 on every pixel): CLAHE turns it into thousands of specks per threshold, and some blob
    # group of SimpleBlobDetector collects more than the 48 centres a group record holds (cpe_dev.h GCAP, OVF_GROUPS = 1024).
    # The oracle's lists are unbounded there; it ends the frame with "no saturated spot".  Every other frame: exact status.
    _compare(cpe, orc, gpu, frames, check_planes=False, allow_overflow={2: 1024})
    det = cpe.api.detect_grid_batch(frames.to(gpu))
    assert int(det['status'][3]) == 0 and int(det['n'][3]) > 0        # the good frame is unaffected by its neighbours


@pytest.mark.gpu
def test_degraded_full_size_frames_beyond_the_round_2_tables(cpe, orc, gpu):
    """tools/stress_parity.py seeds 4011 and 9508 (1920x1200, +-8..9 DN of noise, an intensity ramp, boxes, a dead band): 8 000
    joints, 4 100 / 5 500 of them inside the region rectangle, 220-250 label groups per direction.  Round 2's tables (4096
    joints, 128 groups of 256) ended them with CPE_ST_OVERFLOW while the oracle made 660 / 1220 grid points of them; with
    the capacities of include/cpe.h (one pool of joints shared by the groups) the two sides agree point for point."""
    from cpe_amd import synth
    frames = []
    for seed in (4011, 9508):
        rng = np.random.default_rng(seed)
        b = synth.render_batch(1, 1200, 1920, seed=seed, with_gt=False)
        d, _ = synth.degrade(b['left'][0].numpy(), rng)
        frames.append(d)
    n_ok = _compare(cpe, orc, gpu, torch.from_numpy(np.stack(frames)), check_planes=False)
    assert n_ok == 2


@pytest.mark.gpu
def test_line_stage_capacities_are_the_same_on_both_sides(cpe, orc, gpu):
    """a frame beyond a capacity of include/cpe.h ends with CPE_ST_OVERFLOW in the library and in the oracle alike: pure
    noise over a rendered grid makes tens of thousands of joints"""
    rng = np.random.default_rng(5)
    f = _frames(600, 800, 1, 2).numpy().copy()
    noisy = np.clip(f[0].astype(int) + rng.integers(-40, 41, size=f[0].shape), 0, 255).astype(np.uint8)
    from oracle import stages as S
    ref = S.detect_grid(noisy, debug=True)
    det = cpe.api.detect_grid_batch(torch.from_numpy(noisy[None]).to(gpu))
    assert int(det['status'][0]) == ref['status'], (det['ws'].state()[0], ref['status'], ref['n_joints'], ref['n_cyl_joints'])


@pytest.mark.gpu
def test_reference_signature_and_json(cpe, orc, gpu):
    import json
    from oracle import stages as S
    f = _frames(480, 640, 1, 0).numpy()
    out = cpe.api.detect_grid(f[0])
    assert out is not None and len(out) == 4
    col_img, result_json, rows_updated, cols_updated = out
    assert col_img.shape == (480, 640, 3) and col_img.dtype == np.uint8
    d = json.loads(result_json)
    assert list(d.keys()) == ['center_point', 'points'] and list(d['points'][0].keys()) == ['id', 'x', 'y']
    ref = S.detect_grid(f[0])
    assert [p['id'] for p in d['points']] == ref['id'].tolist()
    assert np.array_equal(np.array([[p['x'], p['y']] for p in d['points']]), ref['xy'])
    assert all(p['id'][0] >= 0 for p in d['points'])                                 # remove_minus_labels
    assert [tuple(p['id']) for p in d['points']] == sorted(tuple(p['id']) for p in d['points'])   # make_json sort
    bgr = np.repeat(f[0][..., None], 3, axis=2)                                       # cv2.imread-style grey BGR
    assert cpe.api.detect_grid(bgr)[1] == result_json
    assert cpe.api.detect_grid(np.full((480, 640), 5, np.uint8)) is None              # reference: prints + None


@pytest.mark.gpu
def test_pipeline_records_and_profile(cpe, orc, gpu):
    """FramePipeline end to end vs oracle (records), chunking invariance, and the per-kernel timers"""
    from oracle import stages as S
    from cpe_amd import synth, pipeline
    b = synth.render_batch(3, 480, 640, seed=5, with_gt=False)
    L, R = b['left'].to(gpu), b['right'].to(gpu)
    p1 = pipeline.FramePipeline(480, 640, b['K1'], b['K2'], b['T21'], 45.0, chunk=3, device=gpu)
    p2 = pipeline.FramePipeline(480, 640, b['K1'], b['K2'], b['T21'], 45.0, chunk=2, device=gpu)
    cpe.lib.profile(True)
    r1 = p1.run(L, R)
    torch.cuda.synchronize()
    rep = cpe.lib.profile_report()
    cpe.lib.profile(False)
    assert any('k_preprocess' in r[0] for r in rep) and any('k_fit_cylinder' in r[0] for r in rep)
    assert all(r[2] >= 0 for r in rep)
    r2 = p2.run(L, R)
    assert torch.equal(r1, r2)                                      # ragged last chunk, same results
    stereo = torch.stack([L, R], 1).contiguous()                    # frame-major pairs: read in place, no concatenation
    assert p2._interleaved(stereo[:, 0], stereo[:, 1]) and not p2._interleaved(L, R)
    for lanes in (1, 2):
        p3 = pipeline.FramePipeline(480, 640, b['K1'], b['K2'], b['T21'], 45.0, chunk=2, device=gpu, lanes=lanes)
        assert torch.equal(r1, p3.run(stereo[:, 0], stereo[:, 1]))
    n_pts, iters, fs, dl, dr = pipeline.unpack_counters(r1[:, 15])
    for i in range(3):
        a, c = S.detect_grid(b['left'][i].numpy()), S.detect_grid(b['right'][i].numpy())
        assert (int(dl[i]), int(dr[i])) == (a['status'], c['status'])
        if a['status'] == 0 and c['status'] == 0:
            ref = orc.fit_single_cylinder(np.concatenate([a['xy'], a['id']], 1), np.concatenate([c['xy'], c['id']], 1),
                                          b['K1'], b['K2'], b['T21'], 45.0)
            assert np.array_equal(r1[i, :12].cpu().numpy(), ref['cyl'].ravel())
            assert np.array_equal(r1[i, 12:14].cpu().numpy(), ref['fvals'])
            assert int(n_pts[i]) == len(ref['pts3']) and int(iters[i]) == ref['iters']


@pytest.mark.gpu
@pytest.mark.parametrize('h,w,seed', [(480, 640, 0), (1200, 1920, 13)])
def test_subpixel_mode_matches_oracle(cpe, orc, gpu, h, w, seed):
    """optional stage f-4 (modify_grayscale_Cline, disabled in the reference's live path): GPU == oracle, bit for bit,
    including the frames on which the reference's code raises (status 7)"""
    from oracle import stages as S
    frames = _frames(h, w, 2, seed)
    det = cpe.api.detect_grid_batch(frames.to(gpu), subpixel=True)
    torch.cuda.synchronize()
    n_ok = 0
    for i in range(frames.shape[0]):
        ref = S.detect_grid(frames[i].numpy(), subpixel=True)
        assert int(det['status'][i]) == ref['status'], i
        if ref['status'] != 0:
            continue
        n_ok += 1
        m = int(det['n'][i])
        assert m == len(ref['xy'])
        assert np.array_equal(det['id'][i, :m].cpu().numpy(), ref['id'])
        assert np.array_equal(det['xy'][i, :m].cpu().numpy(), ref['xy']), np.abs(det['xy'][i, :m].cpu().numpy() - ref['xy']).max()
        base = S.detect_grid(frames[i].numpy())
        assert not np.array_equal(base['xy'], ref['xy'])                  # the stage does move the points
    assert n_ok >= 2


@pytest.mark.gpu
@pytest.mark.parametrize('env', [{'CPE_SERIAL': '1'}, {'CPE_MERGE_REPLAY': '1'}, {'CPE_SERIAL': '1', 'CPE_MERGE_REPLAY': '1'},
                                 {'CPE_MERGE_REPLAY': '2'}, {'CPE_MERGE_REPLAY': '3'}])   # 2: bucketed ranking of every threshold's blobs
def test_execution_variants_give_identical_results(cpe, orc, gpu, env, monkeypatch):
    """stream overlap on / off and the two paths of the batched blob grouping (lane-per-blob vs in-order replay) are
    scheduling choices only: tables, statuses and the order-dependent intermediates must not change by a bit"""
    frames = _frames(600, 800, 3, 21).to(gpu)
    keys = ('xy', 'id', 'n', 'center', 'status')
    base = cpe.api.detect_grid_batch(frames)
    torch.cuda.synchronize()
    ref = {k: base[k].clone() for k in keys}
    ref_planes = {k: base['ws'].plane(k).clone() for k in ('mask_contour', 'exp_h', 'exp_v')}
    ref_state = base['ws'].state()
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    alt = cpe.api.detect_grid_batch(frames)
    torch.cuda.synchronize()
    for k in keys:
        assert torch.equal(alt[k], ref[k]), k
    for k, v in ref_planes.items():
        assert torch.equal(alt['ws'].plane(k), v), k
    st = alt['ws'].state()
    for a, b in zip(st, ref_state):
        for fld in ('n_groups', 'n_kp', 'rect0', 'rect1', 'rect2', 'rect3', 'r0', 'n_joints', 'n_rows', 'n_cols'):
            assert a[fld] == b[fld], fld


@pytest.mark.gpu
def test_full_size_batch_invariance(cpe, orc, gpu):
    """BASELINE-size frames (1920x1200), too many for the oracle in a test: size-independent properties instead.
    Frames are independent, so the pose records may not depend on how the batch is chunked, on the number of chunks in
    flight, or on the order of the frames; and a frame's record equals the one it gets when processed alone."""
    from cpe_amd import synth, pipeline
    F = 24
    b = synth.render_batch(F, 1200, 1920, seed=31, device=gpu, with_gt=False)
    mk = lambda chunk, lanes: pipeline.FramePipeline(1200, 1920, b['K1'], b['K2'], b['T21'], b['radius'], chunk=chunk,
                                                     device=gpu, lanes=lanes)
    ref = mk(24, 1).run(b['left'], b['right'])
    torch.cuda.synchronize()
    assert torch.equal(mk(5, 1).run(b['left'], b['right']), ref)            # ragged chunks
    assert torch.equal(mk(8, 3).run(b['left'], b['right']), ref)            # three chunks in flight
    perm = torch.randperm(F, generator=torch.Generator().manual_seed(1)).to(gpu)
    assert torch.equal(mk(24, 1).run(b['left'][perm].contiguous(), b['right'][perm].contiguous()), ref[perm])
    one = mk(1, 1).run(b['left'][7:8], b['right'][7:8])
    assert torch.equal(one[0], ref[7])
    n_pts, iters, fit_st, dl, dr = pipeline.unpack_counters(ref[:, 15])
    assert int(((fit_st == 0) & (dl == 0) & (dr == 0)).sum()) >= F - 2       # the synthetic frames are detectable


def _paint_lens(g, xc, yc, L, a, dash):
    """two thin bright arcs that meet at both ends (a lens) with a short bright dash in the middle: after the 20x1 opening
    and the 3x3 open / close the lens is ONE line fragment with a hole and the dash is a fragment INSIDE that hole"""
    yy, xx = np.mgrid[:g.shape[0], :g.shape[1]].astype(np.float32)
    t = (xx - xc) / L
    for sgn in (-1, 1):
        d = np.abs(yy - (yc + sgn * a * (1 - t * t)))
        g += np.where(np.abs(t) <= 1, 210 * np.exp(-0.5 * (d / 1.6) ** 2), 0)
    g += np.where(np.abs(xx - xc) <= dash, 210 * np.exp(-0.5 * (np.abs(yy - yc) / 1.6) ** 2), 0)
    return g


@pytest.mark.gpu
def test_fragment_inside_a_hole_of_another_is_dropped(cpe, orc, gpu):
    """RETR_EXTERNAL at expand_line_roi (util_cylinder.py:161): a line fragment that lies in a hole of another fragment is not
    a contour for the reference, so it is neither measured nor expanded.  The frame really contains such a fragment (checked on
    the oracle's masks), and the GPU agrees with the oracle on every stage."""
    from cpe_amd import synth
    from oracle import stages as S
    b = synth.render_batch(1, 1200, 1920, seed=5, with_gt=False)
    r = S.detect_grid(b['left'][0].numpy())
    x0, y0, rw, rh = r['rect']
    g = _paint_lens(b['left'][0].numpy().astype(np.float32), x0 + rw // 4, y0 + rh // 3 + 17, 70, 9, 18)
    img = np.clip(np.round(g), 0, 255).astype(np.uint8)
    ref = S.detect_grid(img, debug=True)
    base = S.close_rect(ref['roi_h'], 3, 3)
    n_all = sum(1 for p, hole in S.find_contours(base, 'list', 'simple') if not hole)
    n_ext = len(S.find_contours(base, 'external', 'simple'))
    assert n_ext < n_all                                    # the nesting exists at this call site
    n_ok = _compare(cpe, orc, gpu, torch.from_numpy(np.stack([img, b['right'][0].numpy()])))
    assert n_ok == 2


@pytest.mark.gpu
def test_no_capacity_overflow_on_clean_4k_frames(cpe, gpu):
    """BASELINE config 5's frame size: 64 clean synthetic 3840x2160 images, none may end in the build-defined status 6.
    (Round 1 lost 8.6 % of them to the 1024-point tables and to 64 joints per label group; a frame whose expanded row
    masks touch puts 4 x 39 joints into one group.)"""
    from cpe_amd import synth
    b = synth.render_batch(32, 2160, 3840, seed=1000, device='cuda', with_gt=False)
    ws = None
    worst = 0
    for part in (b['left'], b['right']):
        for i0 in range(0, 32, 8):
            frames = part[i0:i0 + 8].contiguous()
            if ws is None:
                ws = cpe.api.DetectWorkspace(8, 2160, 3840, frames.device)
            det = cpe.api.detect_grid_batch(frames, ws)
            torch.cuda.synchronize()
            st = det['status'].cpu().tolist()
            assert st == [0] * 8, (i0, st, [s['overflow'] for s in ws.state()])
            worst = max(worst, int(det['n'].max()))
    assert 600 < worst <= cpe.fit.MAXP


@pytest.mark.gpu
def test_noisy_small_frames_fit_the_pooled_sweep_lists(cpe, gpu):
    """The sweep's component lists are pools over the 17 thresholds (cpe_dev.h sweep_pool).  Small frames with heavy sensor
    noise are the densest case per pixel: the seeds of tools/stress_parity.py that overflowed a first, tighter sizing."""
    import importlib.util
    import torch
    from cpe_amd import api, synth
    spec = importlib.util.spec_from_file_location('stress_parity', os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tools', 'stress_parity.py'))
    sp = importlib.util.module_from_spec(spec); spec.loader.exec_module(sp)
    OVF_SWEEP = 1 << 11
    for seed in (1012, 1019, 1038, 1052, 1062, 1064, 1071, 9503):
        rng = np.random.default_rng(seed)
        h, w = (1200, 1920) if seed == 9503 else sp.SIZES[seed % len(sp.SIZES)]   # 9503: noise 11 + ramp at full size, 770 000 dark components
        b = synth.render_batch(1, h, w, seed=seed, with_gt=False)
        frames = np.stack([sp.degrade(img, rng)[0] for img in (b['left'][0].numpy(), b['right'][0].numpy())])
        det = api.detect_grid_batch(torch.from_numpy(frames).to('cuda:0'))
        torch.cuda.synchronize()
        for i, d in enumerate(det['ws'].state()):
            assert not (d['overflow'] & OVF_SWEEP), (seed, i, d['overflow'])


@pytest.mark.gpu
def test_heavy_sensor_noise_at_full_size_is_not_a_capacity_overflow(cpe, gpu):
    """1920x1200 frames with uniform noise of +-7 .. +-11 DN (the frames of tools/overflow_census.py): CLAHE turns that noise
    into tens of thousands of specks per threshold -- up to 24 000 blobs per threshold and 22 000 blob groups.  None of them
    may end as status 6 (they did until the blob / group / component tables were doubled in round 2)."""
    from cpe_amd import api, synth
    b = synth.render_batch(5, 1200, 1920, seed=77, device='cuda', with_gt=False)
    fr = torch.cat([b['left'], b['right']]).cpu().numpy().astype(np.int32)
    rng = np.random.default_rng(5)
    for i in range(fr.shape[0]):
        a = 7 + i % 5
        fr[i] += rng.integers(-a, a + 1, size=fr[i].shape)
    det = api.detect_grid_batch(torch.from_numpy(np.clip(fr, 0, 255).astype(np.uint8)).to(gpu))
    torch.cuda.synchronize()
    st = det['ws'].state()
    assert [int(s) for s in det['status']].count(6) == 0, [(int(s), d['overflow']) for s, d in zip(det['status'], st)]
    assert max(d['n_groups'] for d in st) > 16384 or max(int(v) for v in det['ws'].plane('sweep')[:, 42:59].max(1).values) > 16384
