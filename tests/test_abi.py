"""The C-ABI library loads on a GPU-less host and exports every symbol include/mmn_hip.h declares.
Only host-side helpers are called here (no kernels, no device memory)."""
import ctypes as C
import os
import re

import pytest

import multimodn_amd
from multimodn_amd import hip

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from multimodn_amd import build
    build.build()
    return hip.load()


def declared_symbols():
    text = open(os.path.join(REPO, "include", "mmn_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mmn_[a-z_]+)\s*\(", text)))


def test_header_symbols_all_exported(lib):
    syms = declared_symbols()
    assert len(syms) >= 19
    for s in syms:
        assert hasattr(lib, s), s
    assert set(syms) == set(hip.ABI_SYMBOLS)


def test_version_and_errors(lib):
    assert lib.mmn_version() == hip.VERSION
    assert lib.mmn_error_string(0) == b"ok"
    assert b"workspace" in lib.mmn_error_string(-3)


def test_struct_sizes_match_header():
    assert C.sizeof(hip.Linear) == 40
    assert C.sizeof(hip.Encoder) == 16 + 40 * hip.MAX_LAYERS
    assert C.sizeof(hip.Model) == 32 + C.sizeof(hip.Encoder) * hip.MAX_ENCODERS + 32 * hip.MAX_DECODERS
    assert C.sizeof(hip.Batch) == 8 * 16 + 4 * 16 + 16 + 16 + 64 + 64 + 16      # + tile_rows, tile_seq (ABI 101)


def _model(S=128, F=64, H=(32, 32), E=4, D=3):
    m = hip.Model()
    m.state_size, m.n_encoders, m.n_decoders = S, E, D
    for e in range(E):
        me = m.enc[e]
        me.n_features, me.n_layers, me.activation = F, len(H) + 1, hip.ACT_RELU
        dims = [F, *H]
        for l, (a, b) in enumerate(zip(dims, dims[1:])):
            me.layer[l].in_dim, me.layer[l].out_dim = a, b
        me.layer[len(H)].in_dim, me.layer[len(H)].out_dim = dims[-1] + S, S
    return m


def test_host_side_sizing(lib):
    m = _model()
    R, D, E = 5, 3, 4
    assert lib.mmn_stats_floats(C.byref(m)) == R * D + E + 5 * R * D + R + 4
    assert lib.mmn_epoch_doubles(C.byref(m)) == R * D + E + 5 * R * D + R + 1
    small, big = lib.mmn_workspace_bytes(C.byref(m), 256), lib.mmn_workspace_bytes(C.byref(m), 4096)
    assert 0 < small < big < 1 << 30
    bad = _model()
    bad.enc[0].layer[2].in_dim = 7          # inconsistent with hidden + state
    assert lib.mmn_workspace_bytes(C.byref(bad), 256) == 0


def test_plan_create_rejects_bad_arguments(lib):
    m = _model()
    plan = C.c_void_p()
    assert lib.mmn_plan_create(C.byref(m), 0, None, 0, None, C.byref(plan)) == -1
    assert lib.mmn_plan_create(C.byref(m), 64, 1, 1 << 30, 4096, C.byref(plan)) == -3    # misaligned workspace


def test_missing_library_fails_loudly(tmp_path):
    with pytest.raises(hip.MmnError):
        hip.load(str(tmp_path / "libmmn_hip.so"))
