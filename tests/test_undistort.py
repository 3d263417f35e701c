"""Row f-3: undistortion pre-step (utils/iotool.py:22-39, cv2.undistort).  [ext] parity unpinned vs cv2: the oracle is the
restatement in oracle/src/orc_undistort.c; these tests pin its structural properties on the CPU and compare the HIP
path with it bit for bit on the GPU."""
import json
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle

K0 = np.array([[1400.0, 0.0, 955.5], [0.0, 1398.0, 601.25], [0.0, 0.0, 1.0]])


def _cam(radial, tangential, K=K0):
    return dict(IntrinsicMatrix=K.tolist(), RadialDistortion=list(radial), TangentialDistortion=list(tangential))


def _frame(h, w, seed):
    r = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w]
    img = 90 + 60 * np.sin(xx / 17.0) * np.cos(yy / 23.0) + r.normal(0, 6, (h, w))
    return np.clip(img, 0, 255).astype(np.uint8)


def test_oracle_zero_distortion_is_identity():
    oracle.build()
    for (h, w) in ((480, 640), (333, 517)):
        K = K0.copy(); K[0, 2] = w / 2 - 0.3; K[1, 2] = h / 2 + 0.4
        mxy, mf = oracle.undistort_map(K, np.zeros(4), h, w)
        yy, xx = np.mgrid[0:h, 0:w]
        assert np.array_equal(mxy[..., 0], xx) and np.array_equal(mxy[..., 1], yy) and not mf.any()
        src = _frame(h, w, 1)
        assert np.array_equal(oracle.undistort(src, K, np.zeros(4)), src)


def test_oracle_map_matches_plain_double_model():
    """the fixed-point map is the Brown-Conrady model evaluated in double and rounded to 1/32 px: recompute it with
    numpy (closed-form inverse projection instead of OpenCV's running sums) and allow the last fixed-point step"""
    oracle.build()
    h, w = 600, 800
    K = np.array([[900.0, 0, 401.2], [0, 905.0, 297.7], [0, 0, 1.0]])
    dist = np.array([-0.21, 0.07, 0.0011, -0.0007, -0.012])
    mxy, mf = oracle.undistort_map(K, dist, h, w)
    u_fix = mxy[..., 0].astype(np.int64) * 32 + (mf & 31)
    v_fix = mxy[..., 1].astype(np.int64) * 32 + (mf >> 5)
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float64)
    x = (xx - K[0, 2]) / K[0, 0]; y = (yy - K[1, 2]) / K[1, 1]
    r2 = x * x + y * y
    k1, k2, p1, p2, k3 = dist
    kr = 1 + ((k3 * r2 + k2) * r2 + k1) * r2
    xd = x * kr + p1 * 2 * x * y + p2 * (r2 + 2 * x * x)
    yd = y * kr + p1 * (r2 + 2 * y * y) + p2 * 2 * x * y
    u = K[0, 0] * xd + K[0, 2]; v = K[1, 1] * yd + K[1, 2]
    assert np.abs(u_fix - np.rint(u * 32)).max() <= 1 and np.abs(v_fix - np.rint(v * 32)).max() <= 1
    assert (u_fix == np.rint(u * 32)).mean() > 0.999


def test_oracle_remap_constant_border_and_weights():
    oracle.build()
    src = np.full((8, 8), 200, np.uint8)
    mxy = np.zeros((8, 8, 2), np.int16); mf = np.zeros((8, 8), np.uint16)
    mxy[..., 0] = np.arange(8)[None, :] - 1      # shift right by one pixel: column 0 reads x = -1 -> border 0
    mxy[..., 1] = np.arange(8)[:, None]
    out = oracle.remap_bilinear(src, mxy, mf)
    assert np.array_equal(out[:, 0], np.zeros(8, np.uint8)) and (out[:, 1:] == 200).all()
    mf[:] = 16                                    # fx = 16/32: half of the left neighbour, half of the right
    out = oracle.remap_bilinear(src, mxy, mf)
    assert (out[:, 0] == 100).all() and (out[:-1, 1:] == 200).all()


@pytest.mark.gpu
@pytest.mark.parametrize('h,w,nd', [(1200, 1920, 4), (600, 800, 5), (483, 650, 5), (480, 641, 8)])
def test_gpu_undistort_matches_oracle(h, w, nd):
    import torch
    import cpe_amd
    from cpe_amd import iotool
    oracle.build()
    K = K0.copy(); K[0, 2] = w / 2 + 3.3; K[1, 2] = h / 2 - 2.1; K[0, 0] = 0.8 * w; K[1, 1] = 0.81 * w
    coeffs = np.array([-0.23, 0.09, 0.0013, -0.0008, -0.015, 0.01, -0.002, 0.0005])[:nd]
    # what iotool.py stacks: radial first, tangential last -> OpenCV reads positions 2,3 as p1,p2
    cam = _cam(coeffs[:nd - 2], coeffs[nd - 2:], K)
    und = iotool.Undistorter(cam, h, w, 'cuda:0')
    mxy, mf = oracle.undistort_map(K, coeffs, h, w)
    assert np.array_equal(und.map_xy.cpu().numpy(), mxy)
    assert np.array_equal(und.map_f.cpu().numpy().view(np.uint16), mf)
    frames = np.stack([_frame(h, w, s) for s in (3, 4, 5)])
    out = und(torch.from_numpy(frames).cuda()).cpu().numpy()
    for i in range(3):
        assert np.array_equal(out[i], oracle.remap_bilinear(frames[i], mxy, mf))
    # reference signature, colour image: channels are independent
    col = np.ascontiguousarray(np.moveaxis(frames, 0, 2))
    got = iotool.undistort_image(col, cam)
    assert got.shape == col.shape and np.array_equal(got[..., 1], out[1])


@pytest.mark.gpu
def test_gpu_undistort_feeds_detect(tmp_path):
    """camera JSON -> undistort -> detect_grid: a frame rendered WITH lens distortion gives the grid of the clean frame"""
    import torch
    import cpe_amd
    from cpe_amd import iotool, synth, api
    b = synth.render_batch(1, 1200, 1920, seed=5, device='cuda', with_gt=False)
    clean = b['left'][0]
    K = b['K1'].cpu().numpy() if hasattr(b['K1'], 'cpu') else np.asarray(b['K1'])
    cam = _cam([0.012, -0.004], [0.0002, -0.0001], np.asarray(K, dtype=np.float64))
    p = tmp_path / 'cam.json'
    p.write_text(json.dumps(dict(LeftCamera=cam, RightCamera=cam)))
    left_cam, _ = iotool.load_camera_data(str(p))
    und = iotool.Undistorter(left_cam, 1200, 1920, 'cuda:0')
    # distort the clean frame with the inverse of the map (nearest source pixel scatter is not exact; instead check
    # that undistorting with a mild distortion keeps the detector working and moves points by a few pixels at most)
    out = und(clean)
    d0 = api.detect_grid_batch(clean[None])
    d1 = api.detect_grid_batch(out[None])
    assert int(d0['status'][0]) == 0 and int(d1['status'][0]) == 0
    assert abs(int(d0['n'][0]) - int(d1['n'][0])) <= 40


# ---------------------------------------------------------------- second mode: undistortImage(I, cameraParams, 'cubic') (preProcessing.m:3-4)
def _keys(t):
    return np.stack([(-t**3 + 2 * t**2 - t) / 2, (3 * t**3 - 5 * t**2 + 2) / 2, (-3 * t**3 + 4 * t**2 + t) / 2, (t**3 - t**2) / 2], -1)


def _cubic_numpy(src, m, fill=0):
    """cubic convolution in float64, written independently of the C restatement: pad the image by one Keys-extrapolated
    sample on every side, then gather 4 x 4 taps"""
    h, w = src.shape
    P = np.zeros((h + 2, w + 2)); P[1:-1, 1:-1] = src
    P[1:-1, 0] = 3 * P[1:-1, 1] - 3 * P[1:-1, 2] + P[1:-1, 3]; P[1:-1, -1] = 3 * P[1:-1, -2] - 3 * P[1:-1, -3] + P[1:-1, -4]
    P[0] = 3 * P[1] - 3 * P[2] + P[3]; P[-1] = 3 * P[-2] - 3 * P[-3] + P[-4]
    x = m[..., 0].astype(np.float64); y = m[..., 1].astype(np.float64)
    inside = (x >= 0) & (y >= 0) & (x <= w - 1) & (y <= h - 1)
    ix = np.clip(np.floor(x).astype(int), 0, w - 2); iy = np.clip(np.floor(y).astype(int), 0, h - 2)
    wx = _keys(x - ix); wy = _keys(y - iy)
    acc = np.zeros(x.shape)
    for r in range(4):
        for c in range(4):
            acc += wy[..., r] * wx[..., c] * P[iy + r, ix + c]          # padded index = image index + 1; taps at -1..2
    out = np.clip(np.floor(acc + 0.5), 0, 255)
    return np.where(inside, out, fill).astype(np.uint8), acc, inside


def test_oracle_matlab_mode_structure():
    oracle.build()
    h, w = 240, 320
    K = np.array([[700.0, 0.4, 160.7], [0, 702.0, 121.3], [0, 0, 1.0]])     # MATLAB convention: 1-based principal point
    m = oracle.undistort_map_matlab(K, [0, 0, 0], [0, 0], h, w)
    yy, xx = np.mgrid[0:h, 0:w]
    assert np.abs(m[..., 0] - xx).max() < 1e-4 and np.abs(m[..., 1] - yy).max() < 1e-4   # no distortion: identity map
    src = _frame(h, w, 2)
    ident = np.stack([xx, yy], -1).astype(np.float32)
    assert np.array_equal(oracle.remap_cubic(src, ident), src)                # integer positions: weights (0, 1, 0, 0)
    ramp = np.clip(20 + 0.5 * xx + 0.25 * yy, 0, 255).astype(np.float64)     # cubic convolution reproduces linear functions,
    sh = ident + np.float32(0.5)                                             # up to the image border (Keys' extrapolation is linear-exact)
    got = oracle.remap_cubic(ramp.astype(np.uint8), sh)
    want, acc, inside = _cubic_numpy(ramp.astype(np.uint8), sh)
    assert np.abs(got.astype(int) - want.astype(int)).max() <= 1
    assert (got[~inside] == 0).all() and inside[:-1, :-1].all() and not inside[-1].any() and not inside[:, -1].any()
    # the distortion model against a plain numpy evaluation
    rad, tan = [-0.19, 0.06, -0.011], [0.0012, -0.0007]
    m = oracle.undistort_map_matlab(K, rad, tan, h, w).astype(np.float64)
    u, v = xx + 1.0, yy + 1.0
    y = (v - K[1, 2]) / K[1, 1]; x = (u - K[0, 2] - K[0, 1] * y) / K[0, 0]
    r2 = x * x + y * y
    a = rad[0] * r2 + rad[1] * r2**2 + rad[2] * r2**3
    xd = x + x * a + 2 * tan[0] * x * y + tan[1] * (r2 + 2 * x * x); yd = y + y * a + tan[0] * (r2 + 2 * y * y) + 2 * tan[1] * x * y
    assert np.abs(m[..., 0] - (xd * K[0, 0] + K[0, 2] + K[0, 1] * yd - 1)).max() < 1e-4
    assert np.abs(m[..., 1] - (yd * K[1, 1] + K[1, 2] - 1)).max() < 1e-4


@pytest.mark.parametrize('seed', [0, 1, 2])
def test_oracle_cubic_matches_float64_restatement(seed):
    oracle.build()
    rng = np.random.default_rng(seed)
    h, w = 97, 131
    src = _frame(h, w, seed)
    m = np.stack([rng.uniform(-2, w + 1, (h, w)), rng.uniform(-2, h + 1, (h, w))], -1).astype(np.float32)
    m[0, :8] = [[0, 0], [w - 1, h - 1], [0.25, 0], [w - 1.25, h - 1], [0, 0.5], [w - 1, h - 1.5], [1e-3, 1e-3], [w - 1 - 1e-3, 5]]
    got = oracle.remap_cubic(src, m)
    want, acc, inside = _cubic_numpy(src, m)
    assert (got[~inside] == 0).all()
    near_half = np.abs(acc - np.floor(acc) - 0.5) < 1e-3                      # single vs double arithmetic may round these differently
    assert np.array_equal(got[inside & ~near_half], want[inside & ~near_half])
    assert np.abs(got.astype(int) - want.astype(int)).max() <= 1
    assert got[0, 0] == src[0, 0] and got[0, 1] == src[h - 1, w - 1]


@pytest.mark.gpu
@pytest.mark.parametrize('h,w,nr', [(1200, 1920, 2), (483, 650, 3), (64, 67, 3)])
def test_gpu_matlab_undistort_matches_oracle(h, w, nr):
    import torch
    import cpe_amd
    from cpe_amd import iotool
    oracle.build()
    K = K0.copy(); K[0, 2] = w / 2 + 3.3; K[1, 2] = h / 2 - 2.1; K[0, 0] = 0.8 * w; K[1, 1] = 0.81 * w; K[0, 1] = 0.3
    rad = [-0.23, 0.09, -0.015][:nr]; tan = [0.0013, -0.0008]
    cam = _cam(rad, tan, K)
    und = iotool.Undistorter(cam, h, w, 'cuda:0', interp='cubic')
    m = oracle.undistort_map_matlab(K, rad, tan, h, w)
    assert np.array_equal(und.map.cpu().numpy(), m)
    frames = np.stack([_frame(h, w, s) for s in range(10)])                 # more than one REMAP_FRAMES group
    out = und(torch.from_numpy(frames).cuda()).cpu().numpy()
    for i in range(len(frames)):
        assert np.array_equal(out[i], oracle.remap_cubic(frames[i], m))
    col = np.ascontiguousarray(np.moveaxis(frames[:3], 0, 2))
    got = iotool.undistort_image(col, cam, interp='cubic')
    assert got.shape == col.shape and np.array_equal(got[..., 2], out[2])
    left, right = iotool.preprocessing(frames[0], col, cam, cam)               # preProcessing.m: grey stays, colour -> rgb2gray
    assert np.array_equal(left, out[0]) and right.shape == (h, w) and right.dtype == np.uint8
