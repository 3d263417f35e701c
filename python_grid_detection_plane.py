"""Drop-in entry module with the reference's name and entry points
(python_grid_detection_plane.py:13-70 process_images_in_folder, :74-119 detect_grid), backed by the
MI355X HIP kernels (row f-2: the planar-target variant over utils/util_plane.py; point ids are (row, col) here).
The cylinder script's MATLAB call pattern, for reference:

    py_func = py.importlib.import_module('python_grid_detection_plane');    % makePyGridPts.m:15
    outputs = py_func.detect_grid(py.numpy.array(input_img));                  % makePyGridPts.m:29
    gridPts = jsondecode(char(outputs{2}));                                    % makePyGridPts.m:39-41

Scope: grey or grey-replicated BGR frames.  process_images_in_folder undistorts every image first
(utils/iotool.py:22-39, row f-3: cpe_amd.iotool.undistort_image on the GPU), as the reference's CLI does.
"""
import json
import os
import sys

import numpy as np

_ROOT = os.path.dirname(os.path.abspath(__file__))
if _ROOT not in sys.path:
    sys.path.insert(0, _ROOT)

import cpe_amd  # noqa: E402
from cpe_amd import api as _api  # noqa: E402


def detect_grid(input_img):
    """(col_img, result_json, rows_updated, cols_updated), or None after printing the error, as the reference."""
    try:
        return _api.detect_grid(input_img, target='plane')
    except (TypeError, ValueError, NotImplementedError) as e:     # the reference's blanket try/except (:111-112)
        print(f"Error in detect_grid: {e}")
        return None


def load_camera_data(json_path):
    """utils/iotool.py:8-20"""
    with open(json_path, 'r') as f:
        cam = json.load(f)
    return cam["LeftCamera"], cam["RightCamera"]


def _undistort(img, params):
    from cpe_amd import iotool
    return iotool.undistort_image(img, params)


def process_images_in_folder(json_path, folder_path, output_folder=None):
    """python_grid_detection_plane.py:12-64: every image of the folder -> <stem>_arc<ext> + processed_images_data.json"""
    from PIL import Image
    left_camera_params, right_camera_params = load_camera_data(json_path)
    if output_folder is None:
        output_folder = folder_path
    if not os.path.exists(output_folder):
        os.makedirs(output_folder)
    valid_exts = ('.png', '.jpg', '.jpeg', '.bmp', '.tif', '.tiff')
    image_files = [f for f in os.listdir(folder_path) if f.lower().endswith(valid_exts)]
    if not image_files:
        print(f"No images found in folder: {folder_path}")
        return
    images_json_data = {}
    for filename in image_files:
        img = np.asarray(Image.open(os.path.join(folder_path, filename)).convert('L'))
        if 'L' in filename:
            und = _undistort(img, left_camera_params)
        elif 'R' in filename:
            und = _undistort(img, right_camera_params)
        else:
            raise ValueError(f"Unknown camera type in filename: {filename}")
        out_img, result_json, _, _ = detect_grid(und)             # a failed frame raises here, as the reference's unpack does
        base_name = os.path.splitext(filename)[0]
        try:
            images_json_data[base_name] = json.loads(result_json)
        except json.JSONDecodeError:
            print(f"Invalid JSON data for image {filename}. Skipping.")
            continue
        Image.fromarray(out_img[..., ::-1]).save(os.path.join(output_folder, f"{base_name}_arc{os.path.splitext(filename)[1]}"))
    output_json_path = os.path.join(output_folder, "processed_images_data.json")
    with open(output_json_path, 'w') as json_file:
        json.dump(images_json_data, json_file, indent=4)
    print(f"Data saved to {output_json_path}")
    return json.dumps(images_json_data)


if __name__ == "__main__":
    if len(sys.argv) < 3:
        print('usage: python python_grid_detection_plane.py <stereoParams.json> <input folder> [output folder]')
        sys.exit(2)
    process_images_in_folder(sys.argv[1], sys.argv[2], sys.argv[3] if len(sys.argv) > 3 else None)
