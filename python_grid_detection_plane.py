"""Drop-in entry module of the planar-target script (python_grid_detection_plane.py:13-70 process_images_in_folder,
:74-119 detect_grid; row f-2), backed by the MI355X HIP kernels.  Point ids are (row, col) here, as in the reference.
Everything but the target is shared with python_grid_detection_cylinder.py."""
import os
import sys

_ROOT = os.path.dirname(os.path.abspath(__file__))
if _ROOT not in sys.path:
    sys.path.insert(0, _ROOT)

import cpe_amd  # noqa: E402,F401
from cpe_amd import api as _api, folder as _folder  # noqa: E402

TARGET = 'plane'


def detect_grid(input_img):
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
