"""Folder entry point of both scripts, batched.

The reference walks a folder one file at a time: imread -> cv2.undistort (which rebuilds its map per image) ->
detect_grid -> imwrite, and dumps the decoded JSON of all images at the end (python_grid_detection_cylinder.py:12-64,
python_grid_detection_plane.py:13-70).  Here the folder is walked in windows of a few dozen files: every camera's undistortion map is built
once (iotool.Undistorter), the frames of a window go through the remap kernel and detect_grid_batch together, and their
pictures are written before the next window is read.  Files written and the returned string are the reference's: `<stem>_arc<ext>` per image and
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
    """list of same-size numpy images of one camera -> undistorted frames on the device: u8 [n,h,w] if all are grey, else
    [n,h,w,3] BGR (grey ones replicated).  cv2.undistort works per channel; the conversions of detect_grid come after it."""
    dev = und.device
    if all(a.ndim == 2 for a in images):
        return und(torch.from_numpy(np.stack(images)).to(dev))
    planes = torch.from_numpy(np.stack([np.moveaxis(a, 2, 0) if a.ndim == 3 else np.repeat(a[None], 3, 0) for a in images])).to(dev)   # [n,3,h,w]
    m = planes.shape[0]
    flat = und(planes.reshape(m * 3, und.h, und.w).contiguous()).reshape(m, 3, und.h, und.w)
    return flat.permute(0, 2, 3, 1).contiguous()


def run_folder(json_path, folder_path, output_folder=None, target='cylinder', chunk=32, device='cuda:0'):
    """process_images_in_folder(json_path, folder_path, output_folder=None) -> JSON string (None for an empty folder).

    The folder is walked in listing order in windows of `chunk` files: read -> undistort -> detect -> write, so host and
    device memory are bounded by one window whatever the folder holds, and -- as in the reference's file-by-file loop --
    the `_arc` pictures of the files before a failure are on disk when it is raised: a frame on which detect_grid fails
    stops the run with the TypeError the reference's tuple-unpack of None raises, a file name without L / R with its
    ValueError.  Undistortion maps (one per camera and frame size) and workspaces (one per frame size) are kept across
    windows."""
    cams = dict(zip(('left', 'right'), iotool.load_camera_data(json_path)))
    dst = folder_path if output_folder is None else output_folder
    os.makedirs(dst, exist_ok=True)
    names = [f for f in os.listdir(folder_path) if f.lower().endswith(IMAGE_SUFFIXES)]
    if not names:
        print(f'No images found in folder: {folder_path}')
        return None
    und_cache, ws_cache, collected = {}, {}, {}
    for w0 in range(0, len(names), max(1, int(chunk))):
        window = names[w0:w0 + max(1, int(chunk))]
        keys, bad_name = [], None
        for f in window:
            try:
                keys.append(camera_key(f))
            except ValueError as e:                     # raised when the loop gets there: the files before it are processed
                bad_name = e
                break
        window = window[:len(keys)]
        images = [read_image(os.path.join(folder_path, f)) for f in window]
        results = [None] * len(window)
        groups = {}
        for i, (k, a) in enumerate(zip(keys, images)):
            groups.setdefault((k,) + a.shape[:2], []).append(i)
        for (k, h, w), members in groups.items():
            und = und_cache.get((k, h, w))
            if und is None:
                und = und_cache[(k, h, w)] = iotool.Undistorter(cams[k], h, w, device)
            batch = undistort_group([images[i] for i in members], und)
            ws = ws_cache.get((h, w))
            if ws is None or not ws.fits(len(members), h, w):
                ws = ws_cache[(h, w)] = api.DetectWorkspace(len(members), h, w, batch.device)
            det = api.detect_grid_batch(batch, ws, target=target)
            host = batch.cpu().numpy()
            for j, i in enumerate(members):             # (reads the line tables of this call before the workspace moves on)
                results[i] = api.frame_result(det, j, host[j], target)
        for name, res in zip(window, results):
            if res is None:
                raise TypeError(f'cannot unpack non-iterable NoneType object (detect_grid failed on {name})')
            picture, result_json = res[0], res[1]
            stem, ext = os.path.splitext(name)
            collected[stem] = json.loads(result_json)
            write_image(os.path.join(dst, f'{stem}_arc{ext}'), picture)
        if bad_name is not None:
            raise bad_name
    out_path = os.path.join(dst, 'processed_images_data.json')
    with open(out_path, 'w') as fh:
        json.dump(collected, fh, indent=4)
    print(f'Data saved to {out_path}')
    return json.dumps(collected)
