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


def test_preprocess_kernel_keeps_its_pending_load_registers_untouched():
    """k_preprocess waits for its inline-asm gray loads two intervals after issuing them; the ISA the installed hipcc
    generates must not read those registers in between (tools/check_preprocess_asm.py)"""
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location('check_preprocess_asm', os.path.join(root, 'tools', 'check_preprocess_asm.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    assert mod.check() is None
