"""Row f-3: the undistortion pre-step of the reference's CLI entry point, mirror of utils/iotool.py.

    load_camera_data(json_path)                 iotool.py:8-20   (camera JSON written by createCameraDataJSON.m:7-12)
    undistort_image(image, camera_params)       iotool.py:22-39  (cv2.undistort, bilinear)
    Undistorter(camera_params, h, w, device)    the batched form: the fixed-point map is built once per camera
                                                (cv2.undistort rebuilds it for every image), frames are one gather pass

    Undistorter(..., interp='cubic') /         the MATLAB entry point's pre-step instead: undistortImage(I, cameraParams,
    undistort_image(..., interp='cubic')        'cubic') of utils/preProcessing.m:3-4 (distortPoints map, cubic convolution)

The planar entry module (python_grid_detection_plane.py) and the cylinder one call the bilinear form for every image of a
folder.  No CPU fallback: the HIP library does the work."""
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
    """undistortion map of one camera, resident on the GPU.  interp='linear': cv2.undistort (fixed-point map in the
    CV_16SC2 + CV_16UC1 layout, bilinear); interp='cubic': MATLAB's undistortImage(I, cameraParams, 'cubic') -- the camera
    JSON's matrix read with MATLAB's 1-based principal point, RadialDistortion (2 or 3 terms) and TangentialDistortion kept
    apart, float32 source coordinates, cubic convolution, fill value 0."""

    def __init__(self, camera_params, h, w, device='cuda:0', interp='linear'):
        if interp not in ('linear', 'cubic'):
            raise _lib.CpeError(f"Undistorter: interp must be 'linear' or 'cubic' (got {interp!r})")
        self.h, self.w, self.device, self.interp = int(h), int(w), torch.device(device), interp
        K, dist = camera_arrays(camera_params)
        self.K = np.ascontiguousarray(K)
        L = _lib.load()
        if interp == 'cubic':
            radial = np.ascontiguousarray(np.asarray(camera_params['RadialDistortion'], dtype=np.float64).ravel())
            tang = np.ascontiguousarray(np.asarray(camera_params['TangentialDistortion'], dtype=np.float64).ravel())
            if radial.size not in (2, 3) or tang.size != 2:
                raise _lib.CpeError('MATLAB camera parameters hold 2 or 3 radial and 2 tangential coefficients')
            self.map = torch.empty((self.h, self.w, 2), dtype=torch.float32, device=self.device)
            with torch.cuda.device(self.device):
                _lib.check(L.cpe_undistort_map_matlab(self.K.ctypes.data_as(C.c_void_p), radial.ctypes.data_as(C.c_void_p),
                                                      int(radial.size), tang.ctypes.data_as(C.c_void_p), self.h, self.w,
                                                      self.map.data_ptr(), torch.cuda.current_stream().cuda_stream),
                           'cpe_undistort_map_matlab')
            return
        if dist.size not in (0, 4, 5, 8, 12):
            raise _lib.CpeError(f'{dist.size} distortion coefficients: OpenCV takes 4, 5, 8 or 12')
        self.dist = np.ascontiguousarray(dist)
        self.map_xy = torch.empty((self.h, self.w, 2), dtype=torch.int16, device=self.device)
        self.map_f = torch.empty((self.h, self.w), dtype=torch.int16, device=self.device)   # bit pattern of u16
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
            s = torch.cuda.current_stream().cuda_stream
            if self.interp == 'cubic':
                _lib.check(L.cpe_remap_cubic_batch(f.data_ptr(), f.shape[0], self.h, self.w, self.map.data_ptr(), 0,
                                                   dst.data_ptr(), s), 'cpe_remap_cubic_batch')
            else:
                _lib.check(L.cpe_remap_bilinear_batch(f.data_ptr(), f.shape[0], self.h, self.w, self.map_xy.data_ptr(),
                                                      self.map_f.data_ptr(), dst.data_ptr(), s), 'cpe_remap_bilinear_batch')
        return dst[0] if single else dst


def undistort_image(image, camera_params, device='cuda:0', interp='linear'):
    """reference signature (iotool.py:22): numpy u8 image [h,w] or [h,w,c] -> undistorted numpy image.
    (Channels are independent in cv2.undistort / undistortImage; they are processed as a batch of planes.)
    interp='cubic' = the MATLAB entry point's undistortImage(I, cameraParams, 'cubic') (preProcessing.m:3-4)."""
    img = np.array(image, copy=True, order='C')     # (PIL / MATLAB hand over read-only buffers)
    if img.dtype != np.uint8 or img.ndim not in (2, 3):
        raise _lib.CpeError('undistort_image: u8 image [h,w] or [h,w,c] expected')
    planes = img[None] if img.ndim == 2 else np.ascontiguousarray(np.moveaxis(img, 2, 0))
    und = Undistorter(camera_params, img.shape[0], img.shape[1], device, interp)
    out = und(torch.from_numpy(planes).to(und.device)).cpu().numpy()
    return out[0] if img.ndim == 2 else np.ascontiguousarray(np.moveaxis(out, 0, 2))


def preprocessing(input_img_l, input_img_r, camera_params_l, camera_params_r, device='cuda:0'):
    """utils/preProcessing.m: im2uint8 + undistortImage(..., 'cubic') + rgb2gray for both cameras -> (imgL_uint8, imgR_uint8).
    (Its third and fourth outputs, adapthisteq pictures, feed nothing on the detection path: exp_gridDetection.m:67-68 hands
    the undistorted images to makePyGridPts.)  Colour input is converted AFTER the undistortion, as in the .m file."""
    out = []
    for img, cam in ((input_img_l, camera_params_l), (input_img_r, camera_params_r)):
        u = undistort_image(img, cam, device, 'cubic')
        if u.ndim == 3:   # rgb2gray: the first row of inv([1 .956 .621; 1 -.272 -.647; 1 -1.106 1.703]) on R, G, B, rounded (on the device)
            t = torch.from_numpy(u).to(device).to(torch.float64)
            g = t[..., 0] * 0.298936021293775 + t[..., 1] * 0.587043074451121 + t[..., 2] * 0.114020904255103
            u = torch.floor(g + 0.5).clamp_(0, 255).to(torch.uint8).cpu().numpy()
        out.append(u)
    return out[0], out[1]
