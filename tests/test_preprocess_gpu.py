"""a-1 load_and_preprocess_image: HIP kernel (through the C ABI) vs the oracle -- bit-exact mask."""
import numpy as np
import pytest
import torch


def _run(cpe, dev, frames):
    g = torch.from_numpy(frames).to(dev)
    n, h, w = g.shape
    m = torch.empty_like(g)
    lib = cpe.lib.load()
    cpe.lib.check(lib.cpe_preprocess_batch(g.data_ptr(), n, h, w, m.data_ptr(),
                                           torch.cuda.current_stream().cuda_stream), 'cpe_preprocess_batch')
    torch.cuda.synchronize()
    return m.cpu().numpy()


@pytest.mark.gpu
@pytest.mark.parametrize('h,w', [(480, 640), (97, 131), (64, 64), (33, 200)])
def test_mask_bit_exact_synthetic(cpe, orc, gpu, h, w):
    from cpe_amd import synth
    b = synth.render_batch(2, h, w, seed=3, with_gt=False)
    frames = np.concatenate([b['left'].numpy(), b['right'].numpy()])
    got = _run(cpe, gpu, frames)
    for i in range(frames.shape[0]):
        _, want = orc.preprocess(frames[i])
        assert np.array_equal(got[i], want), f'frame {i}: {(got[i] != want).sum()} px differ'


@pytest.mark.gpu
def test_mask_bit_exact_noise(cpe, orc, gpu):
    rng = np.random.default_rng(0)
    frames = rng.integers(0, 256, size=(3, 150, 210), dtype=np.uint8)
    frames[1] = 0
    frames[2, :, :100] = 255
    got = _run(cpe, gpu, frames)
    for i in range(3):
        _, want = orc.preprocess(frames[i])
        assert np.array_equal(got[i], want)


@pytest.mark.gpu
def test_full_size_properties(cpe, orc, gpu):
    """1920x1200: tile-boundary independence -- any crop far from the border equals the oracle on
    the same crop region computed from the full frame."""
    from cpe_amd import synth
    b = synth.render_batch(1, 1200, 1920, seed=5, device='cuda', with_gt=False)
    f = b['left'].cpu().numpy()
    got = _run(cpe, gpu, f)
    _, want = orc.preprocess(f[0])
    assert np.array_equal(got[0], want)


@pytest.mark.gpu
def test_bad_args(cpe, gpu):
    lib = cpe.lib.load()
    assert lib.cpe_preprocess_batch(None, 1, 64, 64, None, None) < 0
    assert b'null' in lib.cpe_last_error_string()
