"""CPU: the C-ABI library loads and exports every symbol include/deepmod_hip.h declares."""
import os
import re

from conftest import ROOT
from deepmod_amd import _lib


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "deepmod_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(dm_[a-z0-9_]+)\s*\(", text)))


def test_header_and_binding_agree(hip_lib):
    declared = _declared_symbols()
    bound = sorted(n for n, _, _ in _lib.SIGNATURES)
    assert declared == bound
    for name in declared:
        assert getattr(hip_lib, name) is not None


def test_version_and_error_channel(hip_lib):
    assert b"gfx950" in hip_lib.dm_version()
    # bad arguments are rejected before any device work, message via dm_last_error()
    h = hip_lib.dm_model_create(0, None, 0, 7, 100, 21, 3)
    assert not h
    assert b"expected 408402" in hip_lib.dm_last_error()
    assert hip_lib.dm_predict_windows(None, None, 0, None, None) != 0


def test_flatten_matches_oracle_blob():
    import numpy as np
    from deepmod_amd import model, synth
    from oracle import oracle_np
    w = synth.synthetic_weights(1, 1.0)
    assert np.array_equal(model.flatten_weights(w), oracle_np.flatten_weights(w))
    assert model.flatten_weights(w).size == _lib.DM_WEIGHT_FLOATS


def test_header_constants_equal_the_bindings():
    """Every DM_OPT_* / DM_PREC_* / DM_INFO_* / error code the header defines has the same value in deepmod_amd/_lib.py (a constant that
    drifts between the two would select another kernel or misread an error without any symbol going missing)."""
    text = open(os.path.join(ROOT, "include", "deepmod_hip.h")).read()
    defines = {n: int(v) for n, v in re.findall(r"^#define\s+(DM_(?:OPT|PREC|INFO|OK|E[A-Z]+|MAP|ROWS)[A-Z0-9_]*)\s+\(?(-?\d+)\)?", text, flags=re.M)}
    assert {"DM_OPT_PRECISION", "DM_OPT_F16X3_SHAPE", "DM_PREC_F16X3", "DM_PREC_F16I8", "DM_PREC_F16X3_ROLES", "DM_INFO_HAS_F16X3_ROLES", "DM_ERANGE"} <= set(defines)
    checked = 0
    for name, value in defines.items():
        if hasattr(_lib, name):
            assert getattr(_lib, name) == value, name
            checked += 1
    assert checked >= 20
