"""CPU: the C-ABI library builds, loads and exports every symbol include/xm3d.h declares;
host-side argument validation works without a GPU (no compute calls here)."""
import ctypes

import pytest

from xmask3d_amd import _lib


@pytest.fixture(scope="module")
def handle():
    import __graft_entry__ as g

    g.build()
    return _lib.lib()


def test_exports_every_declared_symbol(handle):
    syms = _lib.header_symbols()
    assert len(syms) >= 20
    missing = [s for s in syms if not hasattr(handle, s)]
    assert missing == [], f"declared in include/xm3d.h but not exported: {missing}"
    assert set(_lib._SIGS) == set(syms), set(_lib._SIGS) ^ set(syms)


def test_version_and_error_string(handle):
    assert handle.xm3d_version() >= 100
    assert isinstance(handle.xm3d_last_error(), bytes)


def test_argument_validation_without_gpu(handle):
    need = ctypes.c_size_t(0)
    assert handle.xm3d_voxelize_ws_bytes(1000, ctypes.byref(need)) == 0 and need.value > 1000 * 8
    assert handle.xm3d_voxelize_ws_bytes(-1, ctypes.byref(need)) == -1
    # cin not a multiple of 16 is rejected by the packer before any device work
    assert handle.xm3d_spconv_pack_weight(ctypes.c_void_p(16), 27, 3, 32, ctypes.c_void_p(16), None) == -1
    assert b"multiples of 16" in handle.xm3d_last_error()
    # hash capacity must be a power of two >= 2n
    assert handle.xm3d_hash_build(None, 100, ctypes.c_void_p(16), ctypes.c_void_p(16), 100, None) == -1


def test_product_path_fails_loudly_without_device():
    import torch

    from xmask3d_amd import ops

    if torch.cuda.is_available():
        pytest.skip("checks the no-GPU behaviour")
    with pytest.raises(RuntimeError, match="no CPU path"):
        ops.voxelize(torch.zeros(4, 3, dtype=torch.float64), [[1, 0, 0, 0], [0, 1, 0, 0], [0, 0, 1, 0], [0, 0, 0, 1]])
    with pytest.raises(RuntimeError, match="Not implemented on the CPU"):
        from xmask3d_amd import msda
        msda.ms_deform_attn_forward(torch.zeros(1, 4, 1, 4), torch.tensor([[2, 2]]), torch.tensor([0]),
                                    torch.zeros(1, 1, 1, 1, 1, 2), torch.zeros(1, 1, 1, 1, 1), 64)
