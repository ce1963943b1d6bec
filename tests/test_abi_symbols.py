"""The C-ABI library loads without a GPU and exports every symbol include/vgan_hip.h declares (CPU tier)."""
import ctypes
import os
import re

import pytest

from conftest import REPO


def declared_functions():
    text = open(os.path.join(REPO, "include", "vgan_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(vgan_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_the_expected_surface():
    names = declared_functions()
    for must in ["vgan_linear_forward", "vgan_linear_backward_input", "vgan_linear_backward_params", "vgan_mask_project_forward",
                 "vgan_mask_backward", "vgan_colmax", "vgan_mmd_build_tiles", "vgan_mmd_gram", "vgan_mmd_reduce", "vgan_mmd_loss",
                 "vgan_mmd_backward", "vgan_adadelta_step", "vgan_noise_normal"]:
        assert must in names


def test_library_exports_every_declared_symbol():
    import vgan_amd
    lib_path = vgan_amd.lib.LIB_PATH
    if not os.path.exists(lib_path):
        pytest.fail(f"{lib_path} missing: run `python -c 'import __graft_entry__ as g; g.build()'` first")
    lib = ctypes.CDLL(lib_path)
    for name in declared_functions():
        assert hasattr(lib, name), f"{name} declared in include/vgan_hip.h but not exported"
    assert lib.vgan_abi_version() == vgan_amd.lib.ABI_VERSION


def test_binding_table_matches_header():
    import vgan_amd
    assert sorted(vgan_amd.lib.SIGNATURES) == declared_functions()
    vgan_amd.lib.load()  # sets argtypes for every entry; raises on a missing symbol


def test_product_has_no_cpu_fallback_and_never_imports_the_oracle():
    import torch
    import vgan_amd
    from vgan_amd.ops import HipOps
    if not torch.cuda.is_available():
        with pytest.raises(vgan_amd.lib.VganHipError):
            HipOps()
    pkg = os.path.join(REPO, "v-gan_amd")
    for root, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(root, f)).read()
                assert "oracle" not in re.sub(r'""".*?"""', "", src, flags=re.S).replace("# ", ""), f"{f} mentions the oracle"


def test_argument_validation_needs_no_gpu():
    """Every entry point validates its arguments before it touches the device: bad calls return VGAN_ERR_ARG (1) and leave
    a message in vgan_last_error(), with or without a GPU."""
    import vgan_amd
    lib = vgan_amd.lib.load()
    null = None
    assert lib.vgan_linear_forward(null, 0, 1, 0, null, 0, null, null, 0, 0, 0, 0, null) != 0
    assert b"bad argument" in lib.vgan_last_error()
    assert lib.vgan_mmd_gram(null, 0, null, 0, 0, null, null, 0, 0, null, 0, 0, null, null) != 0
    assert lib.vgan_mmd_gram_bf3(null, null, 0, null, 0, null, null, 0, 64, null, null, 0, 0, null, null, 0, 0, 0, null, 0, 0, null, 0, null, 0, null) != 0
    assert lib.vgan_mask_project_forward_bf3(null, 0, null, 0, null, null, 1, 0, null, null, 0, null, null, null, 0, null, null, 0, 0, 0,
                                             null, 1, null, null, null) != 0
    assert lib.vgan_col_mean(null, 0, 0, 0, null, null) != 0
    assert lib.vgan_gemm_grouped(null, 0, null) != 0
    assert lib.vgan_adadelta_step(null, null, 1, 0, null, null, 0, 0.1, 0.9, 1e-6, 0.0, 1.0, null) != 0
    assert lib.vgan_rbf_kernel_matrix(null, 0, 0, 0, null, 1.0, null, 0, null) != 0
    assert b"twosample.hip" in lib.vgan_last_error()
