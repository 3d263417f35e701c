"""The C-ABI library loads and exports every symbol include/cpe.h declares (no compute calls)."""
import ctypes
import os


def test_exports(cpe):
    import cpe_amd
    so = cpe_amd.lib.SO_PATH
    assert os.path.exists(so), 'libcpe_hip.so missing: run __graft_entry__.build()'
    lib = ctypes.CDLL(so)
    names = cpe_amd.lib.declared_symbols()
    assert 'cpe_preprocess_batch' in names
    for n in names:
        assert hasattr(lib, n), f'{n} declared in include/cpe.h but not exported'
    assert cpe_amd.lib.load().cpe_version() >= 100
    assert set(cpe_amd.lib._SIGS) == set(names)

