"""Host-side boundary logic that needs no GPU: the .mat writer, the JSON shape, the folder driver's file conventions."""
import json

import numpy as np
import pytest
import torch


def test_save_mat_structs_round_trip(cpe, tmp_path):
    from scipy.io import loadmat
    left = [dict(center_point=np.array([10.5, 20.25]), points=np.array([[1.5, 2.5, 0, -1], [3.0, 4.0, 0, 0]])),
            dict(center_point=np.array([1.0, 2.0]), points=np.zeros((0, 4)))]
    right = [dict(center_point=np.array([11.5, 21.25]), points=np.array([[5.5, 6.5, 1, 2]])),
             dict(center_point=np.array([3.0, 4.0]), points=np.array([[7.0, 8.0, 2, 2]]))]
    fits = dict(m=torch.tensor([2, 1], dtype=torch.int32), pts3=torch.arange(2 * 4 * 3, dtype=torch.float64).reshape(2, 4, 3),
                cyl=torch.arange(24, dtype=torch.float64).reshape(2, 2, 6), T=torch.eye(4, dtype=torch.float64).reshape(1, 16).repeat(2, 1),
                fvals=torch.tensor([[102.1, 1.05], [3.0, 0.5]], dtype=torch.float64), mean_err=torch.tensor([0.12, 0.2], dtype=torch.float64),
                status=torch.tensor([0, 5], dtype=torch.int32))
    m = loadmat(cpe.api.save_mat(str(tmp_path / 'a.mat'), left, right, fits, names=['00', '-1-4']))
    assert m['gridPtsPair'].shape == (2, 2)
    assert np.array_equal(m['gridPtsPair'][0, 0]['points'], left[0]['points'])
    assert m['gridPtsPair'][1, 0]['points'].shape == (0, 4)
    assert m['gridPtsPair'][0, 1]['center_point'].shape == (2, 1)
    assert m['frames'][0, 0]['pts3'].shape == (3, 2) and m['frames'][0, 1]['pts3'].shape == (3, 1)
    assert np.array_equal(m['frames'][0, 1]['cylParams'], np.arange(12, 24).reshape(2, 6))
    assert np.array_equal(m['frames'][0, 0]['cylT'], np.eye(4))
    assert m['frames'][0, 0]['fvals'].shape == (1, 2)
    with pytest.raises(ValueError):
        cpe.api.save_mat(str(tmp_path / 'b.mat'), left, right[:1])


def test_make_json_shape(cpe):
    """the string pointsStruct2mat.m decodes: fields id, x, y in this order; indent 4"""
    s = cpe.api.make_json(np.array([3.0, 4.0]), np.array([[1.0, 0.5], [1.5, 2.5]]), np.array([[0, -1], [0, 1]]))
    d = json.loads(s)
    assert list(d) == ['center_point', 'points'] and [list(p) for p in d['points']] == [['id', 'x', 'y']] * 2
    assert d['points'][0] == {'id': [0, -1], 'x': 1.0, 'y': 0.5}
    assert s.startswith('{\n    "center_point": [')


def test_folder_conventions(cpe, tmp_path):
    from PIL import Image
    from cpe_amd import folder
    assert folder.camera_key('-1-4L.png') == 'left' and folder.camera_key('00R.png') == 'right'
    assert folder.camera_key('LR.png') == 'left'                 # 'L' is tested first (python_grid_detection_cylinder.py:36)
    with pytest.raises(ValueError):
        folder.camera_key('frame7.png')
    g = (np.arange(48, dtype=np.uint8) * 5).reshape(6, 8)
    Image.fromarray(g).save(tmp_path / 'g.png')
    Image.fromarray(np.repeat(g[..., None], 3, 2)).save(tmp_path / 'g3.png')
    rgb = np.stack([g, g // 2, g // 3], 2)
    Image.fromarray(rgb).save(tmp_path / 'c.png')
    assert np.array_equal(folder.read_image(str(tmp_path / 'g.png')), g)
    assert np.array_equal(folder.read_image(str(tmp_path / 'g3.png')), g)          # imread's replicated channels: one plane
    assert np.array_equal(folder.read_image(str(tmp_path / 'c.png')), rgb[..., ::-1])   # BGR, like cv2.imread
    with pytest.raises(ValueError):
        cpe.api.frames_to_device([np.zeros((2, 2, 4), np.uint8)], 'cpu')
    with pytest.raises(TypeError):
        cpe.api.frames_to_device([np.zeros((2, 2), np.float32)], 'cpu')


def test_oracle_bgr2gray_fixed_point(orc):
    from oracle import stages as S
    rng = np.random.default_rng(0)
    bgr = rng.integers(0, 256, (9, 11, 3), dtype=np.uint8)
    want = ((bgr[..., 0].astype(np.int64) * 3735 + bgr[..., 1].astype(np.int64) * 19235 + bgr[..., 2].astype(np.int64) * 9798 + 16384) >> 15)
    assert np.array_equal(S.bgr2gray(bgr), want.astype(np.uint8))
    g = rng.integers(0, 256, (5, 5), dtype=np.uint8)
    assert np.array_equal(S.bgr2gray(np.repeat(g[..., None], 3, 2)), g)               # identity on grey-replicated frames
