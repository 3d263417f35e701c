"""The oracle against golden vectors produced by the REAL reference functions
(tools/gen_golden.py, generated in the authoring container from /root/reference)."""
import json
import os

import numpy as np

from conftest import GOLDEN


def test_ridges_bit_identical_to_skimage(orc):
    """util_cylinder.detect_ridges (real skimage/scipy) vs oracle: bit-for-bit."""
    z = np.load(os.path.join(GOLDEN, 'ridges.npz'))
    for k in 'abcn':
        emax, emin = orc.detect_ridges(z['img_' + k])
        assert np.array_equal(emin, z['emin_' + k]), k
        assert np.array_equal(emax, z['emax_' + k]), k


def test_intersections_vs_scipy_hybr(orc):
    """poly_intersection_solver (real scipy MINPACK hybr) vs the oracle's Newton restatement:
    same accept/reject decision and solution within 1e-6 px (BASELINE: grid points within 1e-3 px)."""
    from oracle import stages as S
    d = json.load(open(os.path.join(GOLDEN, 'intersections.json')))
    n_sol = 0; worst = 0.0; mism = []
    for c in d['cases']:
        got = S.poly_intersection(c['row'], c['col'])
        want = c['sol']
        if (got is None) != (want is None):
            mism.append((c['row'], c['col'], got, want))
            continue
        if want is not None:
            n_sol += 1
            worst = max(worst, abs(got[0] - want[0]), abs(got[1] - want[1]))
    assert not mism, f'{len(mism)} decision mismatches, first: {mism[0]}'
    assert n_sol > 150
    assert worst < 1e-6, worst


def _labels_from(case, key, H, W):
    lab = np.zeros((H - case['y_off'], W - case['x_off']), np.int32)
    for x, y, v in case[key]:
        lab[y, x] = v
    return lab


def test_topology_vs_reference(orc):
    """group_points_by_label -> create_dummy_rows_cols -> fit_and_draw_polynomial(2) -> remove_label ->
    find_and_assign_intersections_P -> clean_and_relabel, each step against the real functions' output."""
    from oracle import stages as S
    d = json.load(open(os.path.join(GOLDEN, 'topology.json')))
    for case in d['cases']:
        H, W = case['H'], case['W']
        rows = S.group_points(case['centroids'], _labels_from(case, 'lab_h', H, W), case['x_off'], case['y_off'])
        cols = S.group_points(case['centroids'], _labels_from(case, 'lab_v', H, W), case['x_off'], case['y_off'])
        for ls, want in ((rows, case['grouped']['rows']), (cols, case['grouped']['cols'])):
            assert ls.nlines == len(want)
            for g, (lab, pts) in enumerate(want):
                assert ls.label[g] == lab
                assert [(int(a), int(b)) for a, b in ls.points()[g]] == [tuple(p) for p in pts]
        S.fit_lines(cols, False); S.fit_lines(rows, True)
        for ls, want, pre in ((rows, case['fitted']['rows'], 'row'), (cols, case['fitted']['cols'], 'col')):
            for g in range(ls.nlines):
                w_eq = want[f'{pre}{g + 1}']
                np.testing.assert_allclose(ls.equations()[g], w_eq, rtol=1e-9, atol=1e-9)
        S.remove_label(rows, cols)
        assert rows.nlines == len(case['removed']['rows']) and cols.nlines == len(case['removed']['cols'])
        for g in range(rows.nlines):
            np.testing.assert_allclose(rows.equations()[g], case['removed']['rows_eq'][f'col{g + 1}'], rtol=1e-9, atol=1e-9)
        for g in range(cols.nlines):
            np.testing.assert_allclose(cols.equations()[g], case['removed']['cols_eq'][f'col{g + 1}'], rtol=1e-9, atol=1e-9)
        S.intersections(rows, cols, (0, 0, W, H))
        for ls, want in ((rows, case['inter']['rows']), (cols, case['inter']['cols'])):
            for g in range(ls.nlines):
                w_pts = want[f'col{g + 1}']
                got = ls.points()[g]
                assert len(got) == len(w_pts)
                if w_pts:
                    np.testing.assert_allclose(got, w_pts, rtol=0, atol=1e-6)
        S.clean_and_relabel(rows, cols)
        for ls, want, weq, pre in ((rows, case['clean']['rows'], case['clean']['rows_eq'], 'row'),
                                   (cols, case['clean']['cols'], case['clean']['cols_eq'], 'col')):
            assert ls.nlines == len(want)
            for g in range(ls.nlines):
                np.testing.assert_allclose(ls.points()[g], want[f'{pre}{g + 1}'], rtol=0, atol=1e-6)
                np.testing.assert_allclose(ls.equations()[g], weq[f'{pre}{g + 1}'], rtol=1e-9, atol=1e-9)


def test_pca_endpoints_vs_numpy_lapack(orc):
    """get_pca_endpoints (np.cov + np.linalg.eig = LAPACK dgeev) vs the dlanv2 restatement."""
    from oracle import stages as S
    d = json.load(open(os.path.join(GOLDEN, 'pca_endpoints.json')))
    for c in d['cases']:
        p1, p2 = S.pca_endpoints(c['pts'])
        assert list(map(float, p1)) == c['p1'] and list(map(float, p2)) == c['p2']


def test_subpixel_refinement_vs_reference(orc):
    """modify_grayscale_Cline (real function, 2-D grey input, draw_points=False) vs the oracle restatement:
    refitted equations within 1e-9 relative (coefficients of magnitude >= 1e-6; 1e-12 absolute below), domains equal;
    and the case in which the reference raises is reported as status 7."""
    from oracle import stages as S
    z = np.load(os.path.join(GOLDEN, 'subpixel.npz'))
    spec = json.loads(str(z['spec'])); gray = z['gray']
    rk, ck = list(spec['rows_in']), list(spec['cols_in'])
    rows = S.lineset_from_equations([spec['rows_in'][k] for k in rk])
    cols = S.lineset_from_equations([spec['cols_in'][k] for k in ck])
    assert S.subpixel_refine(gray, rows, cols, spec['window'], spec['step']) == 0
    for ls, keys, out in ((rows, rk, spec['rows_out']), (cols, ck, spec['cols_out'])):
        for g, k in enumerate(keys):
            got, want = np.array(ls.equations()[g]), np.array(out[k])
            np.testing.assert_allclose(got[:3], want[:3], rtol=1e-9, atol=1e-12)
            assert np.array_equal(got[3:], want[3:]), k                     # float32-rounded domain, bit for bit
    assert spec['bad_row_raises']
    bad = S.lineset_from_equations([spec['bad_row']]); none = S.lineset_from_equations([])
    assert S.subpixel_refine(gray, bad, none, spec['window'], spec['step']) == 7


def test_plane_lines_vs_reference(orc):
    """row f-2: the planar script's line logic against the real util_plane functions (tests/golden/plane_lines.json):
    degree-1 fits with the merging of short columns, intersections, clean_and_relabel without sorting"""
    from oracle import stages as S
    d = json.load(open(os.path.join(GOLDEN, 'plane_lines.json')))
    n_merged = 0
    for case in d['cases']:
        H, W = case['H'], case['W']
        rows = S.group_points(case['centroids'], _labels_from(case, 'lab_h', H, W), case['x_off'], case['y_off'])
        cols = S.group_points(case['centroids'], _labels_from(case, 'lab_v', H, W), case['x_off'], case['y_off'])
        n_before = cols.nlines
        S.fit_lines_plane(rows, cols)
        want_c, want_r = case['fitted']['cols'], case['fitted']['rows']
        assert cols.nlines == len(want_c) and rows.nlines == len(want_r), (case['seed'], cols.nlines, len(want_c))
        n_merged += n_before - cols.nlines
        for g in range(cols.nlines):
            np.testing.assert_allclose(cols.equations()[g][:5], want_c[f'col{g + 1}'], rtol=1e-9, atol=1e-8)
            got = [(a, b) for a, b in cols.points()[g]]
            assert got == [tuple(p) for p in case['fitted']['col_points'][f'col{g + 1}']]
        for g in range(rows.nlines):
            np.testing.assert_allclose(rows.equations()[g][:5], want_r[f'row{g + 1}'], rtol=1e-9, atol=1e-8)
        S.intersections_plane(rows, cols, (0, 0, W, H))
        for ls, want, pre in ((rows, case['inter']['rows'], 'row'), (cols, case['inter']['cols'], 'col')):
            for g in range(ls.nlines):
                w_pts = want[f'{pre}{g + 1}']
                g_pts = ls.points()[g]
                assert len(g_pts) == len(w_pts), (case['seed'], pre, g)
                if w_pts:
                    assert np.abs(np.array(g_pts) - np.array(w_pts)).max() < 1e-6
        S.clean_plane(rows, cols)
        for ls, want, pre in ((rows, case['clean']['rows'], 'row'), (cols, case['clean']['cols'], 'col')):
            assert ls.nlines == len(want)
            for g in range(ls.nlines):
                assert np.abs(np.array(ls.points()[g]) - np.array(want[f'{pre}{g + 1}'])).max() < 1e-6
    assert n_merged >= 6      # the merge path was exercised


def _lineset_from_points(S, lines):
    ls = S.LineSet(); ls.nlines = len(lines)
    for g, pts in enumerate(lines):
        ls.npts[g] = len(pts)
        for k, (x, y) in enumerate(pts):
            ls.pts[g][k][0] = x; ls.pts[g][k][1] = y
    return ls


def test_indexing_vs_reference(orc):
    """indexing_data (+ remove_minus_labels) + make_json of both scripts against the real functions, the blurred image
    being an input of the fixture (tests/golden/indexing.npz): centre, ids, order, coordinates"""
    import ctypes as C
    from oracle import stages as S
    z = np.load(os.path.join(GOLDEN, 'indexing.npz'))
    spec = json.loads(str(z['spec']))
    for i, case in enumerate(spec['cases']):
        img = z[f'img_{i}']
        rows = _lineset_from_points(S, case['rows']); cols = _lineset_from_points(S, case['cols'])
        if case['variant'] == 'cylinder':
            n, center, xy, ids = S.index_points(rows, cols, img, case['r0'])
        else:
            h, w = img.shape
            center = np.zeros(2); xy = np.zeros((4096, 2)); ids = np.zeros((4096, 2), np.int32)
            n = orc.lib().orc_index_points_plane(C.byref(rows), C.byref(cols), img.ctypes.data_as(C.c_void_p), h, w, case['r0'],
                                                 center.ctypes.data_as(C.c_void_p), xy.ctypes.data_as(C.c_void_p),
                                                 ids.ctypes.data_as(C.c_void_p), 4096)
            xy, ids = xy[:n], ids[:n]
        js = case['json']
        tag = (case['variant'], case['seed'])
        assert n == len(js['points']), tag
        assert list(center) == js['center_point'], tag
        assert ids.tolist() == [p['id'] for p in js['points']], tag
        assert xy.tolist() == [[p['x'], p['y']] for p in js['points']], tag
