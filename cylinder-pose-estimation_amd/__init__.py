"""cpe_amd -- MI355X-native laser-grid detection + cylinder fit (hot path of
cv3vpl-lab/cylinder-pose-estimation).  Python is plumbing (device memory, streams,
torch.distributed); the arithmetic is hand-written HIP behind the C ABI of include/cpe.h."""
from . import lib  # noqa: F401
from . import fit, api, synth, pipeline, dist, multiframe, iotool, folder  # noqa: F401

__all__ = ['lib', 'fit', 'api', 'synth', 'pipeline', 'dist', 'multiframe', 'iotool', 'folder']
