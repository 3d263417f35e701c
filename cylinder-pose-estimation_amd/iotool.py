"""Row f-3: the undistortion pre-step of the reference's CLI entry point, mirror of utils/iotool.py.

    load_camera_data(json_path)                 iotool.py:8-20   (camera JSON written by createCameraDataJSON.m:7-12)
    undistort_image(image, camera_params)       iotool.py:22-39  (cv2.undistort, bilinear)
    Undistorter(camera_params, h, w, device)    the batched form: the fixed-point map is built once per camera
                                                (cv2.undistort rebuilds it for every image), frames are one gather pass

The planar entry module (python_grid_detection_plane.py) and the cylinder one call it for every image of a folder.
The MATLAB entry point undistorts with `undistortImage(..., 'cubic')` (utils/preProcessing.m:3-4) instead; that second
interpolation mode is not built.  No CPU fallback: the HIP library does the work."""
import ctypes as C
import json

import numpy as np
import torch

from . import lib as _lib


def load_camera_data(json_path):
    """-> (LeftCamera, RightCamera) dicts with IntrinsicMatrix, RadialDistortion, TangentialDistortion"""
    with open(json_path, 'r') as f:
        camera_data = json.load(f)
    return camera_data['LeftCamera'], camera_data['RightCamera']


def camera_arrays(camera_params):
    """K (3x3 f64) and the coefficient vector exactly as iotool.py:33-36 hands them to OpenCV:
    hstack((RadialDistortion, TangentialDistortion)) -- read by OpenCV as (k1, k2, p1, p2[, k3]) whatever it holds"""
    K = np.array(camera_params['IntrinsicMatrix'], dtype=np.float64).reshape(3, 3)
    dist = np.hstack((np.asarray(camera_params['RadialDistortion'], dtype=np.float64).ravel(),
                      np.asarray(camera_params['TangentialDistortion'], dtype=np.float64).ravel()))
    return K, dist


class Undistorter:
    """fixed-point undistortion map of one camera (CV_16SC2 + CV_16UC1 layout), resident on the GPU"""

    def __init__(self, camera_params, h, w, device='cuda:0'):
        self.h, self.w, self.device = int(h), int(w), torch.device(device)
        K, dist = camera_arrays(camera_params)
        if dist.size not in (0, 4, 5, 8, 12):
            raise _lib.CpeError(f'{dist.size} distortion coefficients: OpenCV takes 4, 5, 8 or 12')
        self.K, self.dist = np.ascontiguousarray(K), np.ascontiguousarray(dist)
        self.map_xy = torch.empty((self.h, self.w, 2), dtype=torch.int16, device=self.device)
        self.map_f = torch.empty((self.h, self.w), dtype=torch.int16, device=self.device)   # bit pattern of u16
        L = _lib.load()
        with torch.cuda.device(self.device):
            _lib.check(L.cpe_undistort_map(self.K.ctypes.data_as(C.c_void_p), self.dist.ctypes.data_as(C.c_void_p),
                                           int(self.dist.size), self.h, self.w, self.map_xy.data_ptr(), self.map_f.data_ptr(),
                                           torch.cuda.current_stream().cuda_stream), 'cpe_undistort_map')

    def __call__(self, frames, out=None):
        """frames u8 [n,h,w] (or [h,w]) on the device -> undistorted frames, same shape"""
        single = frames.dim() == 2
        f = frames.unsqueeze(0) if single else frames
        if f.dtype != torch.uint8 or f.shape[1:] != (self.h, self.w) or f.device != self.device or not f.is_contiguous():
            raise _lib.CpeError('Undistorter: frames must be contiguous u8 [n,h,w] on the map\'s device')
        dst = torch.empty_like(f) if out is None else out
        L = _lib.load()
        with torch.cuda.device(self.device):
            _lib.check(L.cpe_remap_bilinear_batch(f.data_ptr(), f.shape[0], self.h, self.w, self.map_xy.data_ptr(),
                                                  self.map_f.data_ptr(), dst.data_ptr(),
                                                  torch.cuda.current_stream().cuda_stream), 'cpe_remap_bilinear_batch')
        return dst[0] if single else dst


def undistort_image(image, camera_params, device='cuda:0'):
    """reference signature (iotool.py:22): numpy u8 image [h,w] or [h,w,c] -> undistorted numpy image.
    (Channels are independent in cv2.undistort; they are processed as a batch of planes.)"""
    img = np.array(image, copy=True, order='C')     # (PIL / MATLAB hand over read-only buffers)
    if img.dtype != np.uint8 or img.ndim not in (2, 3):
        raise _lib.CpeError('undistort_image: u8 image [h,w] or [h,w,c] expected')
    planes = img[None] if img.ndim == 2 else np.ascontiguousarray(np.moveaxis(img, 2, 0))
    und = Undistorter(camera_params, img.shape[0], img.shape[1], device)
    out = und(torch.from_numpy(planes).to(und.device)).cpu().numpy()
    return out[0] if img.ndim == 2 else np.ascontiguousarray(np.moveaxis(out, 0, 2))
