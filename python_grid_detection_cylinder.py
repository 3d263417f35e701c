"""Drop-in entry module with the reference's name and entry points
(python_grid_detection_cylinder.py:12-64 process_images_in_folder, :68-112 detect_grid), backed by the
MI355X HIP kernels.  MATLAB keeps calling it unchanged:

    py_func = py.importlib.import_module('python_grid_detection_cylinder');    % makePyGridPts.m:15
    outputs = py_func.detect_grid(py.numpy.array(input_img));                  % makePyGridPts.m:29
    gridPts = jsondecode(char(outputs{2}));                                    % makePyGridPts.m:39-41

Input: 2-D grey frames (what MATLAB passes) or H x W x 3 BGR (what the CLI's imread gives).  The folder driver is the
batched one of cpe_amd/folder.py: maps built once per camera, frames detected in chunks.
"""
import os
import sys

_ROOT = os.path.dirname(os.path.abspath(__file__))
if _ROOT not in sys.path:
    sys.path.insert(0, _ROOT)

import cpe_amd  # noqa: E402,F401
from cpe_amd import api as _api, folder as _folder  # noqa: E402

TARGET = 'cylinder'


def detect_grid(input_img):
    """(col_img, result_json, rows_updated, cols_updated), or None after printing the error: the reference wraps the whole
    body in try / except Exception (:111-112), so nothing propagates to MATLAB"""
    try:
        return _api.detect_grid(input_img, target=TARGET)
    except Exception as e:
        print(f"Error in detect_grid: {e}")
        return None


def process_images_in_folder(json_path, folder_path, output_folder=None):
    return _folder.run_folder(json_path, folder_path, output_folder, target=TARGET)


if __name__ == "__main__":
    if len(sys.argv) < 3:
        print(f'usage: python {os.path.basename(__file__)} <stereoParams.json> <input folder> [output folder]')
        sys.exit(2)
    process_images_in_folder(sys.argv[1], sys.argv[2], sys.argv[3] if len(sys.argv) > 3 else None)
