#!/opt/conda/bin/python3.9
"""Generate golden vectors from the REAL reference functions (this container only).

Run:  /opt/conda/bin/python3.9 -W ignore tools/gen_golden.py

The reference's Python half imports `cv2`, which does not exist in this image.  As recorded in
SURVEY.md section 8c / Appendix A, an EMPTY stub module is registered for `cv2` (only the two
drawing calls `line`/`circle` are no-ops so that drawing code paths pass); no image-processing
function is faked, so only the cv2-free reference functions can run -- those are the ones
exercised here.  One exception, stated where it happens (gen_indexing): indexing_data's single
cv2.GaussianBlur call is the identity there, the blurred image being an INPUT of that fixture.  Library versions are stored in every fixture's metadata.

Outputs (small, committed): tests/golden/*.npz, tests/golden/*.json
Nothing from /root/reference is copied: fixtures are inputs + the outputs the reference produced.
"""
import sys, types, json, os
sys.dont_write_bytecode = True
sys.path.insert(0, '/root/reference')
cv2 = types.ModuleType('cv2')
cv2.line = cv2.circle = lambda *a, **k: None
sys.modules['cv2'] = cv2

import numpy as np
import scipy, skimage
import utils.util_cylinder as uc  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'tests', 'golden')
os.makedirs(OUT, exist_ok=True)
META = dict(python=sys.version.split()[0], numpy=np.__version__, scipy=scipy.__version__,
            skimage=skimage.__version__,
            note='reference pins numpy 1.24.4 / scipy 1.9.3 / scikit-image 0.19.3; generated with the versions above')


def synth_u8(h, w, seed):
    """small laser-grid-like u8 test image (bright lines on dark noisy background)"""
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float64)
    img = 12 + 3 * rng.standard_normal((h, w))
    for k in range(-2, 8):
        d = (yy - (9 + 11.5 * k + 0.04 * xx + 0.0006 * (xx - w / 2) ** 2))
        img += 190 * np.exp(-0.5 * (d / 1.5) ** 2)
        d = (xx - (7 + 12.5 * k + 0.05 * yy))
        img += 170 * np.exp(-0.5 * (d / 1.4) ** 2)
    img[h // 2 - 4:h // 2 + 5, w // 2 - 4:w // 2 + 5] = 255
    return np.clip(np.rint(img), 0, 255).astype(np.uint8)


def gen_ridges():
    cases = {}
    for name, (h, w, seed) in dict(a=(40, 56, 1), b=(64, 48, 2), c=(27, 31, 3)).items():
        img = synth_u8(h, w, seed)
        emax, emin = uc.detect_ridges(img, sigma=3.0)          # util_cylinder.py:1734-1738
        cases['img_' + name] = img
        cases['emax_' + name] = np.asarray(emax, dtype=np.float64)
        cases['emin_' + name] = np.asarray(emin, dtype=np.float64)
    rng = np.random.default_rng(7)
    img = rng.integers(0, 256, size=(33, 45), dtype=np.uint8)  # pure noise, exercises every border
    emax, emin = uc.detect_ridges(img, sigma=3.0)
    cases['img_n'] = img
    cases['emax_n'] = emax
    cases['emin_n'] = emin
    np.savez_compressed(os.path.join(OUT, 'ridges.npz'), meta=json.dumps(META), **cases)


def gen_intersections():
    rng = np.random.default_rng(11)
    vec = []
    # the SURVEY appendix-A example first
    rows = [[1e-4, -0.02, 300, 100, 500, 400]]
    cols = [[-2e-4, 0.05, 250, 200, 400, 200]]
    for _ in range(40):
        a2 = rng.uniform(-3e-4, 3e-4); a1 = rng.uniform(-0.15, 0.15); a0 = rng.uniform(50, 900)
        x0 = rng.uniform(0, 800); x1 = x0 + rng.uniform(100, 900)
        rows.append([a2, a1, a0, x0, x1, abs(x1 - x0)])
        b2 = rng.uniform(-3e-4, 3e-4); b1 = rng.uniform(-0.15, 0.15); b0 = rng.uniform(50, 1500)
        y0 = rng.uniform(0, 500); y1 = y0 + rng.uniform(100, 700)
        cols.append([b2, b1, b0, y0, y1, abs(y1 - y0)])
    rows.append([0, 0, 0, 0, 0, 0]); cols.append([0, 0, 0, 0, 0, 0])     # the [0]*6 dummy equations
    for r in rows:
        for c in cols:
            sol = uc.poly_intersection_solver(r, c, 2)              # util_cylinder.py:1074-1104
            vec.append(dict(row=r, col=c, sol=None if sol is None else [float(sol[0]), float(sol[1])]))
    with open(os.path.join(OUT, 'intersections.json'), 'w') as f:
        json.dump(dict(meta=META, cases=vec), f)


def make_joint_grid(seed, nr, nc, jitter=True):
    """integer joints of a curved grid + labels images so that group_points_by_label can be exercised"""
    rng = np.random.default_rng(seed)
    pts = []
    for r in range(nr):
        for c in range(nc):
            x = 60 + 31.0 * c + 0.8 * r + 0.012 * (r - nr / 2) ** 2
            y = 50 + 29.0 * r + 0.5 * c + 0.02 * (c - nc / 2) ** 2
            if jitter:
                x += rng.uniform(-0.6, 0.6); y += rng.uniform(-0.6, 0.6)
            pts.append((int(x), int(y), r, c))
    return pts


def gen_topology():
    """group_points_by_label -> create_dummy_rows_cols -> fit_and_draw_polynomial -> remove_label ->
    find_and_assign_intersections_P -> clean_and_relabel  (util_cylinder.py:376-550, 1106-1269)"""
    cases = []
    for seed, nr, nc, drop in [(1, 7, 9, 0), (2, 9, 6, 5), (3, 5, 5, 3), (4, 12, 11, 17)]:
        rng = np.random.default_rng(100 + seed)
        pts = make_joint_grid(seed, nr, nc)
        keep = np.ones(len(pts), bool)
        if drop:
            keep[rng.choice(len(pts), size=drop, replace=False)] = False
        pts = [p for p, k in zip(pts, keep) if k]
        order = rng.permutation(len(pts))                   # contour order is arbitrary
        pts = [pts[i] for i in order]
        H, W = 480, 640
        x_off, y_off = 20, 10
        lab_h = np.zeros((H - y_off, W - x_off), np.int32)
        lab_v = np.zeros((H - y_off, W - x_off), np.int32)
        perm_r = rng.permutation(nr) + 1                    # label VALUES are arbitrary
        perm_c = rng.permutation(nc) + 1
        for (x, y, r, c) in pts:
            lab_h[y - y_off, x - x_off] = perm_r[r]
            lab_v[y - y_off, x - x_off] = perm_c[c]
        centroids = [(x, y) for (x, y, r, c) in pts]
        rows = uc.group_points_by_label(centroids, lab_h, x_off, y_off)
        cols = uc.group_points_by_label(centroids, lab_v, x_off, y_off)
        grouped = dict(rows=[[int(l), [list(map(int, p)) for p in ps]] for l, ps in rows],
                       cols=[[int(l), [list(map(int, p)) for p in ps]] for l, ps in cols])
        rows_d, cols_d = uc.create_dummy_rows_cols(rows, cols, degree=2)
        img = np.zeros((H, W, 3), np.uint8)
        _, rows_d, cols_d = uc.fit_and_draw_polynomial(img, rows_d, cols_d, W, H, None, degree=2)
        fitted = dict(rows={k: [float(v) for v in e] for k, e in rows_d['equations'].items()},
                      cols={k: [float(v) for v in e] for k, e in cols_d['equations'].items()})
        rows_d, cols_d = uc.remove_label(rows_d, cols_d)
        removed = dict(rows=list(rows_d['equations'].keys()), cols=list(cols_d['equations'].keys()),
                       rows_eq={k: [float(v) for v in e] for k, e in rows_d['equations'].items()},
                       cols_eq={k: [float(v) for v in e] for k, e in cols_d['equations'].items()})
        _, ru, cu = uc.find_and_assign_intersections_P(img, rows_d, cols_d, None, draw_points=False, degree=2)
        inter = dict(rows={k: [[float(a), float(b)] for a, b in v] for k, v in ru['points'].items()},
                     cols={k: [[float(a), float(b)] for a, b in v] for k, v in cu['points'].items()})
        ru, cu = uc.clean_and_relabel(ru, cu)
        clean = dict(rows={k: [[float(a), float(b)] for a, b in v] for k, v in ru['points'].items()},
                     cols={k: [[float(a), float(b)] for a, b in v] for k, v in cu['points'].items()},
                     rows_eq={k: [float(x) for x in v] for k, v in ru['equations'].items()},
                     cols_eq={k: [float(x) for x in v] for k, v in cu['equations'].items()})
        cases.append(dict(seed=seed, H=H, W=W, x_off=x_off, y_off=y_off,
                          centroids=[list(map(int, p)) for p in centroids],
                          lab_h=[[int(x - x_off), int(y - y_off), int(perm_r[r])] for (x, y, r, c) in pts],
                          lab_v=[[int(x - x_off), int(y - y_off), int(perm_c[c])] for (x, y, r, c) in pts],
                          grouped=grouped, fitted=fitted, removed=removed, inter=inter, clean=clean))
    with open(os.path.join(OUT, 'topology.json'), 'w') as f:
        json.dump(dict(meta=META, cases=cases), f)


def gen_json():
    cases = []
    cd = {"col0": [{"id": (0, 1), "x": 1.5, "y": 2.5}, {"id": (0, -1), "x": 1.0, "y": 0.5}],
          "col-1": [{"id": (-1, 0), "x": 0.5, "y": 1.5}],
          "col2": [{"id": (2, -3), "x": 7.25, "y": 0.125}, {"id": (2, 10), "x": 7.5, "y": 30.0}]}
    kept = uc.remove_minus_labels(cd)                                 # util_cylinder.py:1657-1669
    s = uc.make_json((3.5, 4.25), kept)                               # util_cylinder.py:1674-1727
    cases.append(dict(center=[3.5, 4.25],
                      cols={k: [dict(id=list(p['id']), x=p['x'], y=p['y']) for p in v] for k, v in cd.items()},
                      kept=list(kept.keys()), json=s))
    s2 = uc.make_json((3, 4), {"col0": [{"id": (0, 1), "x": 1.5, "y": 2.5}, {"id": (0, -1), "x": 1.0, "y": 0.5}]})
    cases.append(dict(center=[3, 4], cols={"col0": [dict(id=[0, 1], x=1.5, y=2.5), dict(id=[0, -1], x=1.0, y=0.5)]},
                      kept=["col0"], json=s2))
    with open(os.path.join(OUT, 'make_json.json'), 'w') as f:
        json.dump(dict(meta=META, cases=cases), f)


def gen_pca():
    rng = np.random.default_rng(5)
    cases = []
    for n in (5, 9, 40, 200):
        t = np.sort(rng.uniform(0, 120, n))
        pts = np.stack([10 + t + rng.uniform(-1, 1, n), 300 + 0.07 * t + rng.uniform(-2, 2, n)], 1)
        pts = np.rint(pts).astype(np.float32)
        (x1, y1), (x2, y2) = uc.get_pca_endpoints(pts)                # util_cylinder.py:35-55
        cases.append(dict(pts=pts.tolist(), p1=[float(x1), float(y1)], p2=[float(x2), float(y2)]))
    # vertical-ish
    t = np.arange(30, dtype=np.float32)
    pts = np.stack([50 + np.round(0.1 * t), 20 + t], 1).astype(np.float32)
    (x1, y1), (x2, y2) = uc.get_pca_endpoints(pts)
    cases.append(dict(pts=pts.tolist(), p1=[float(x1), float(y1)], p2=[float(x2), float(y2)]))
    with open(os.path.join(OUT, 'pca_endpoints.json'), 'w') as f:
        json.dump(dict(meta=META, cases=cases), f)


def gen_subpixel():
    """modify_grayscale_Cline (util_cylinder.py:907-971; dead in the live path, call commented at :2040) on a 2-D
    grey image with draw_points=False -- cv2-free, so the REAL function runs: rows y=f(x) are re-fitted after a
    grey-level centre-of-gravity correction of y (compute_center_of_gravity_y :706-751), cols likewise in x."""
    h, w = 200, 260
    rng = np.random.default_rng(21)
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float64)
    img = 10 + 2 * rng.standard_normal((h, w))
    row_defs, col_defs = [], []
    for k in range(5):
        a2, a1, a0 = 0.00008 * (k - 2), 0.03 * (k - 2) + 0.01, 31 + 33.5 * k
        img += 180 * np.exp(-0.5 * ((yy - (a2 * xx ** 2 + a1 * xx + a0)) / 1.6) ** 2)
        row_defs.append((a2, a1, a0))
    for k in range(6):
        b2, b1, b0 = -0.0001 * (k - 2), 0.02 * (k - 3), 26 + 40.25 * k
        img += 170 * np.exp(-0.5 * ((xx - (b2 * yy ** 2 + b1 * yy + b0)) / 1.5) ** 2)
        col_defs.append((b2, b1, b0))
    gray = np.clip(np.rint(img), 0, 255).astype(np.uint8)
    rows = {"points": {}, "equations": {}}
    cols = {"points": {}, "equations": {}}
    for i, (a2, a1, a0) in enumerate(row_defs, 1):     # start equations: the true lines, perturbed by a fraction of a pixel
        x0, x1 = float(rng.uniform(-30, 20)), float(rng.uniform(w - 30, w + 40))
        rows["equations"][f"col{i}"] = [a2 * (1 + 0.02 * rng.standard_normal()), a1 + 1e-3 * rng.standard_normal(),
                                        a0 + rng.uniform(-0.8, 0.8), x0, x1, abs(x1 - x0)]
        rows["points"][f"col{i}"] = []
    for i, (b2, b1, b0) in enumerate(col_defs, 1):
        y0, y1 = float(rng.uniform(-25, 15)), float(rng.uniform(h - 20, h + 30))
        cols["equations"][f"col{i}"] = [b2 * (1 + 0.02 * rng.standard_normal()), b1 + 1e-3 * rng.standard_normal(),
                                        b0 + rng.uniform(-0.8, 0.8), y0, y1, abs(y1 - y0)]
        cols["points"][f"col{i}"] = []
    rows["equations"]["col6"] = [0, 0, 0, 0, 0, 0]; rows["points"]["col6"] = []       # a dummy [0]*6 equation
    _, r2, c2 = uc.modify_grayscale_Cline(gray, rows, cols, draw_points=False, degree=2, sample_step=1.0, window_size=7)
    # a row that leaves the image through the top edge: the reference raises (negative slice stop, :721-741)
    bad = {"points": {"col1": []}, "equations": {"col1": [0.0, -0.2, 20.0, 0.0, 250.0, 250.0]}}
    raised = False
    try:
        uc.modify_grayscale_Cline(gray, bad, {"points": {}, "equations": {}}, draw_points=False, degree=2, sample_step=1.0, window_size=7)
    except ValueError:
        raised = True
    out = dict(meta=META, h=h, w=w, window=7, step=1.0, bad_row=bad["equations"]["col1"], bad_row_raises=raised,
               rows_in={k: [float(x) for x in v] for k, v in rows["equations"].items()},
               cols_in={k: [float(x) for x in v] for k, v in cols["equations"].items()},
               rows_out={k: [float(x) for x in v] for k, v in r2["equations"].items()},
               cols_out={k: [float(x) for x in v] for k, v in c2["equations"].items()})
    np.savez_compressed(os.path.join(OUT, 'subpixel.npz'), gray=gray, spec=json.dumps(out))


def gen_plane_lines():
    """row f-2, the cv2-free line logic of the PLANAR script (utils/util_plane.py): group_points_by_label ->
    create_dummy_rows_cols(1) -> fit_and_draw_polynomial(degree=1, merges runs of short columns, :411-634) ->
    find_and_assign_intersections_P(degree=1) -> clean_and_relabel (:1204-1253: no sorting)"""
    import utils.util_plane as up
    cases = []
    #      seed rows cols  split columns {col: row where the label changes}   joints removed   lonely labels
    specs = [(11, 8, 9, {}, 0, 0), (12, 9, 8, {2: 4, 3: 5}, 3, 0), (13, 10, 10, {0: 3, 1: 3, 2: 6, 7: 5, 8: 5, 9: 2}, 6, 2),
             (14, 6, 7, {4: 2}, 0, 1), (15, 12, 6, {1: 6, 2: 6, 3: 6}, 9, 0)]
    for seed, nr, nc, splits, drop, lonely in specs:
        rng = np.random.default_rng(300 + seed)
        pts = make_joint_grid(seed, nr, nc)
        keep = np.ones(len(pts), bool)
        if drop:
            keep[rng.choice(len(pts), size=drop, replace=False)] = False
        pts = [p for p, k in zip(pts, keep) if k]
        pts = [pts[i] for i in rng.permutation(len(pts))]
        H, W = 480, 640
        x_off, y_off = 20, 10
        lab_h = np.zeros((H - y_off, W - x_off), np.int32)
        lab_v = np.zeros((H - y_off, W - x_off), np.int32)
        perm_r = rng.permutation(nr) + 1
        perm_c = rng.permutation(2 * nc + lonely) + 1      # a split column gets two label values
        centroids = []
        for (x, y, r, c) in pts:
            lab_h[y - y_off, x - x_off] = perm_r[r]
            lv = perm_c[c] if (c not in splits or r < splits[c]) else perm_c[nc + c]
            lab_v[y - y_off, x - x_off] = lv
            centroids.append((x, y))
        for q in range(lonely):                              # a label with a single joint: its column keeps the dummy equation
            x, y = 40 + 7 * q, 300 + 5 * q
            lab_v[y - y_off, x - x_off] = perm_c[2 * nc + q]
            lab_h[y - y_off, x - x_off] = 0
            centroids.append((x, y))
        rows = up.group_points_by_label(centroids, lab_h, x_off, y_off)
        cols = up.group_points_by_label(centroids, lab_v, x_off, y_off)
        rows_d, cols_d = up.create_dummy_rows_cols(rows, cols, degree=1)
        img = np.zeros((H, W, 3), np.uint8)
        _, rows_d, cols_d = up.fit_and_draw_polynomial(img, rows_d, cols_d, W, H, None, degree=1)
        fitted = dict(rows={k: [float(v) for v in e] for k, e in rows_d['equations'].items()},
                      cols={k: [float(v) for v in e] for k, e in cols_d['equations'].items()},
                      col_points={k: [[float(a), float(b)] for a, b in v] for k, v in cols_d['points'].items()})
        _, ru, cu = up.find_and_assign_intersections_P(img, rows_d, cols_d, None, draw_points=False, degree=1)
        inter = dict(rows={k: [[float(a), float(b)] for a, b in v] for k, v in ru['points'].items()},
                     cols={k: [[float(a), float(b)] for a, b in v] for k, v in cu['points'].items()})
        ru, cu = up.clean_and_relabel(ru, cu)
        clean = dict(rows={k: [[float(a), float(b)] for a, b in v] for k, v in ru['points'].items()},
                     cols={k: [[float(a), float(b)] for a, b in v] for k, v in cu['points'].items()})
        cases.append(dict(seed=seed, H=H, W=W, x_off=x_off, y_off=y_off, centroids=[list(map(int, p)) for p in centroids],
                          lab_h=[[int(x - x_off), int(y - y_off), int(lab_h[y - y_off, x - x_off])] for (x, y) in centroids],
                          lab_v=[[int(x - x_off), int(y - y_off), int(lab_v[y - y_off, x - x_off])] for (x, y) in centroids],
                          fitted=fitted, inter=inter, clean=clean))
    with open(os.path.join(OUT, 'plane_lines.json'), 'w') as f:
        json.dump(dict(meta=META, cases=cases), f)


def gen_indexing():
    """indexing_data (+ remove_minus_labels) + make_json of BOTH scripts (util_cylinder.py:1350-1571, util_plane.py:1255-1472).
    The only cv2 call inside is GaussianBlur(input_image, (7,7), 0); here the ALREADY BLURRED image is the fixture's input
    and cv2.GaussianBlur is the identity while the real function runs (the blur itself is [ext] and checked elsewhere), so
    what is pinned is the function's own logic: window means (incl. the plane's half = int(r/4.5), which can be 0),
    first-maximum centre, nearest row / column, ids, ordering."""
    import utils.util_plane as up
    cases = []
    imgs = {}
    for variant, mod in (('cylinder', uc), ('plane', up)):
        for seed, nr, nc, r0 in [(21, 7, 8, 14), (22, 9, 6, 37), (23, 5, 11, 60), (24, 6, 6, 3)]:
            rng = np.random.default_rng(500 + seed)
            H, W = 420, 520
            yy0, xx0 = np.mgrid[0:H, 0:W]
            img = (12 + (xx0 * 7 + yy0 * 13 + seed) % 17).astype(np.uint8)      # deterministic texture (compresses well)
            rows = {"points": {}, "equations": {}}
            cols = {"points": {}, "equations": {}}
            grid = {}
            for r in range(nr):
                for c in range(nc):
                    x = 60.0 + 48.5 * c + 1.7 * r + rng.uniform(-0.4, 0.4)
                    y = 55.0 + 44.25 * r + 0.9 * c + rng.uniform(-0.4, 0.4)
                    if rng.uniform() < 0.08:
                        continue                      # a missing intersection
                    grid[(r, c)] = (float(x), float(y))
            for r in range(nr):
                pts = [grid[(r, c)] for c in range(nc) if (r, c) in grid]
                if pts:
                    rows["points"][f"row{r + 1}"] = pts
            for c in range(nc):
                pts = [grid[(r, c)] for r in range(nr) if (r, c) in grid]
                if pts:
                    cols["points"][f"col{c + 1}"] = pts
            # a bright patch near one grid point decides the centre
            cr, cc = nr // 2, nc // 2 - 1
            while (cr, cc) not in grid:
                cc += 1
            bx, by = grid[(cr, cc)]
            yy, xx = np.mgrid[0:H, 0:W]
            img = np.clip(img + 200 * np.exp(-0.5 * (((xx - bx) / 9.0) ** 2 + ((yy - by) / 9.0) ** 2)), 0, 255).astype(np.uint8)
            saved = getattr(cv2, 'GaussianBlur', None)
            cv2.GaussianBlur = lambda im, k, s_: im
            try:
                out = mod.indexing_data(rows, cols, img, None, r0)
            finally:
                if saved is None:
                    del cv2.GaussianBlur
                else:
                    cv2.GaussianBlur = saved
            result_json, result_dict, rows_dict, cols_dict, center_point = out
            if variant == 'cylinder':
                cols_dict = mod.remove_minus_labels(cols_dict)
            js = json.loads(mod.make_json(center_point, cols_dict))
            imgs[f'img_{len(cases)}'] = img
            cases.append(dict(variant=variant, seed=seed, r0=r0, H=H, W=W,
                              rows=[[list(p) for p in v] for v in rows["points"].values()],
                              cols=[[list(p) for p in v] for v in cols["points"].values()], json=js))
    np.savez_compressed(os.path.join(OUT, 'indexing.npz'), spec=json.dumps(dict(meta=META, cases=cases)), **imgs)


if __name__ == '__main__':
    gen_ridges()
    gen_intersections()
    gen_topology()
    gen_json()
    gen_pca()
    gen_subpixel()
    gen_plane_lines()
    gen_indexing()
    print('golden vectors written to', os.path.abspath(OUT))
