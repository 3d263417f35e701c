"""Folder entry point of both scripts, batched.

The reference walks a folder one file at a time: imread -> cv2.undistort (which rebuilds its map per image) ->
detect_grid -> imwrite, and dumps the decoded JSON of all images at the end (python_grid_detection_cylinder.py:12-64,
python_grid_detection_plane.py:13-70).  Here the folder is read once, every camera's undistortion map is built once
(iotool.Undistorter), the frames of a camera go through the remap kernel together and detect_grid_batch takes them in
chunks.  Files written and the returned string are the reference's: `<stem>_arc<ext>` per image and
`processed_images_data.json` = {stem: decoded result JSON}, indent 4."""
import json
import os

import numpy as np
import torch

from . import api, iotool

IMAGE_SUFFIXES = ('.png', '.jpg', '.jpeg', '.bmp', '.tif', '.tiff')      # :21 of both scripts


def camera_key(filename):
    """file-name convention <pan><tilt>L.png / ...R.png (getUniqueName.m:11-14): 'L' is tested first, as in the reference"""
    if 'L' in filename:
        return 'left'
    if 'R' in filename:
        return 'right'
    raise ValueError(f'Unknown camera type in filename: {filename}')


def read_image(path):
    """decoded like cv2.imread would hand it to undistort: grey files as one plane (imread's three copies are identical, so
    one plane carries the same information), colour files as H x W x 3 BGR"""
    from PIL import Image
    with Image.open(path) as im:
        if im.mode in ('L', '1'):
            return np.asarray(im.convert('L'))
        rgb = np.asarray(im.convert('RGB'))
    if np.array_equal(rgb[..., 0], rgb[..., 1]) and np.array_equal(rgb[..., 1], rgb[..., 2]):
        return np.ascontiguousarray(rgb[..., 0])
    return np.ascontiguousarray(rgb[..., ::-1])


def write_image(path, bgr):
    from PIL import Image
    Image.fromarray(np.ascontiguousarray(bgr[..., ::-1])).save(path)


def undistort_group(images, und):
    """list of same-size numpy images of one camera -> u8 tensor [n,h,w] of undistorted grey frames on the device.
    Colour frames are remapped plane by plane and only then converted (cv2.undistort works per channel and
    load_and_preprocess_image's BGR2GRAY comes after it)."""
    dev = und.device
    out = torch.empty((len(images), und.h, und.w), dtype=torch.uint8, device=dev)
    mono = [i for i, a in enumerate(images) if a.ndim == 2]
    col = [i for i, a in enumerate(images) if a.ndim == 3]
    if mono:
        out[mono] = und(torch.from_numpy(np.stack([images[i] for i in mono])).to(dev))
    if col:
        planes = torch.from_numpy(np.stack([np.moveaxis(images[i], 2, 0) for i in col])).to(dev)      # [m,3,h,w]
        m = planes.shape[0]
        flat = und(planes.reshape(m * 3, und.h, und.w).contiguous()).reshape(m, 3, und.h, und.w)
        out[col] = api.bgr_to_gray(flat.permute(0, 2, 3, 1).contiguous())
    return out


def run_folder(json_path, folder_path, output_folder=None, target='cylinder', chunk=32, device='cuda:0'):
    """process_images_in_folder(json_path, folder_path, output_folder=None) -> JSON string (None for an empty folder).
    A frame on which detect_grid fails stops the run with the TypeError the reference's tuple-unpack of None raises."""
    cams = dict(zip(('left', 'right'), iotool.load_camera_data(json_path)))
    dst = folder_path if output_folder is None else output_folder
    os.makedirs(dst, exist_ok=True)
    names = [f for f in os.listdir(folder_path) if f.lower().endswith(IMAGE_SUFFIXES)]
    if not names:
        print(f'No images found in folder: {folder_path}')
        return None
    keys = [camera_key(f) for f in names]
    images = [read_image(os.path.join(folder_path, f)) for f in names]
    # one undistortion map per (camera, frame size); one remap call per group
    groups = {}
    for i, (k, a) in enumerate(zip(keys, images)):
        groups.setdefault((k,) + a.shape[:2], []).append(i)
    frames = [None] * len(names)
    for (k, h, w), members in groups.items():
        und = iotool.Undistorter(cams[k], h, w, device)
        g = undistort_group([images[i] for i in members], und)
        for j, i in enumerate(members):
            frames[i] = g[j]
    # detect in chunks of equal-size frames (listing order inside a size class)
    results = [None] * len(names)
    by_size = {}
    for i, f in enumerate(frames):
        by_size.setdefault(tuple(f.shape), []).append(i)
    for shape, members in by_size.items():
        ws = None
        for c0 in range(0, len(members), chunk):
            part = members[c0:c0 + chunk]
            batch = torch.stack([frames[i] for i in part])
            if ws is None or not ws.fits(len(part), shape[0], shape[1]):
                ws = api.DetectWorkspace(len(part), shape[0], shape[1], batch.device)
            det = api.detect_grid_batch(batch, ws, target=target)
            host = batch.cpu().numpy()
            for j, i in enumerate(part):
                results[i] = api.frame_result(det, j, host[j], target)
    collected = {}
    for name, res in zip(names, results):
        if res is None:
            raise TypeError(f'cannot unpack non-iterable NoneType object (detect_grid failed on {name})')
        picture, result_json = res[0], res[1]
        stem, ext = os.path.splitext(name)
        collected[stem] = json.loads(result_json)
        write_image(os.path.join(dst, f'{stem}_arc{ext}'), picture)
    out_path = os.path.join(dst, 'processed_images_data.json')
    with open(out_path, 'w') as fh:
        json.dump(collected, fh, indent=4)
    print(f'Data saved to {out_path}')
    return json.dumps(collected)
