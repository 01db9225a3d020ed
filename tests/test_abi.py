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


def test_struct_sizes_match_header(tmp_path):
    """sizeof / offsetof as gcc sees include/mmn_hip.h against the ctypes mirrors in multimodn_amd/hip.py."""
    import subprocess
    src = tmp_path / "sz.c"
    src.write_text(
        '#include <stdio.h>\n#include <stddef.h>\n#include "mmn_hip.h"\n'
        'int main(void){printf("%zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu\\n", sizeof(mmn_linear), sizeof(mmn_encoder),'
        ' sizeof(mmn_decoder), sizeof(mmn_model), sizeof(mmn_batch), sizeof(mmn_adam), offsetof(mmn_batch, drop_mask),'
        ' offsetof(mmn_model, dec), offsetof(mmn_decoder, hidden), sizeof(mmn_step_opts), offsetof(mmn_step_opts, accumulate_epoch),'
        ' offsetof(mmn_batch, flags_ready));'
        'printf("%zu %zu\\n", offsetof(mmn_step_opts, next_drop_p), offsetof(mmn_step_opts, next_drop_floats));return 0;}\n')
    exe = tmp_path / "sz"
    subprocess.run(["gcc", "-I", os.path.join(REPO, "include"), str(src), "-o", str(exe)], check=True)
    got = [int(v) for v in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split()]
    want = [C.sizeof(hip.Linear), C.sizeof(hip.Encoder), C.sizeof(hip.Decoder), C.sizeof(hip.Model), C.sizeof(hip.Batch),
            C.sizeof(hip.AdamDesc), hip.Batch.drop_mask.offset, hip.Model.dec.offset, hip.Decoder.hidden.offset,
            C.sizeof(hip.StepOpts), hip.StepOpts.accumulate_epoch.offset, hip.Batch.flags_ready.offset,
            hip.StepOpts.next_drop_p.offset, hip.StepOpts.next_drop_floats.offset]       # (ABI 105)
    assert got == want
    assert C.sizeof(hip.Linear) == 40 and C.sizeof(hip.Encoder) == 16 + 40 * hip.MAX_LAYERS


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
    assert lib.mmn_stats_floats(C.byref(m)) == R * D + E + 5 * R * D + R + 4 + 2 * hip.MAX_ENCODERS   # + the two NaN-flag sets
    assert lib.mmn_epoch_doubles(C.byref(m)) == R * D + E + 5 * R * D + R + 1
    small, big = lib.mmn_workspace_bytes(C.byref(m), 256), lib.mmn_workspace_bytes(C.byref(m), 4096)
    assert 0 < small < big < 1 << 30
    bad = _model()
    bad.enc[0].layer[2].in_dim = 7          # inconsistent with hidden + state
    assert lib.mmn_workspace_bytes(C.byref(bad), 256) == 0


def test_mimic_family_sizing_and_validation(lib):
    """MIMIC_MLPEncoder / MLPDecoder descriptors (ABI 102): layer 0 reads cat[x, state]; decoder hidden chains from S."""
    S, F, H = 128, 64, (32, 32)
    m = hip.Model()
    m.state_size, m.n_encoders, m.n_decoders = S, 4, 3
    for e in range(4):
        me = m.enc[e]
        me.n_features, me.n_layers, me.activation, me.kind = F, 3, hip.ACT_RELU, hip.ENC_MIMIC
        dims = [F + S, *H, S]
        for l, (a, b) in enumerate(zip(dims, dims[1:])):
            me.layer[l].in_dim, me.layer[l].out_dim = a, b
    for d in range(3):
        md = m.dec[d]
        md.n_hidden, md.hidden_activation = 2, hip.ACT_RELU
        md.hidden[0].in_dim, md.hidden[0].out_dim = S, 32
        md.hidden[1].in_dim, md.hidden[1].out_dim = 32, 32
    plain = lib.mmn_workspace_bytes(C.byref(_model()), 4096)
    ws = lib.mmn_workspace_bytes(C.byref(m), 4096)
    assert ws > plain > 0                                   # + xin / decoder hidden activations and their gradients
    m.enc[1].layer[0].in_dim = F                            # a MIMIC first layer must read F + S columns
    assert lib.mmn_workspace_bytes(C.byref(m), 4096) == 0
    m.enc[1].layer[0].in_dim = F + S
    m.dec[2].hidden[1].in_dim = 31                          # broken hidden chain
    assert lib.mmn_workspace_bytes(C.byref(m), 4096) == 0
    m.dec[2].hidden[1].in_dim = 32
    m.dec[0].n_hidden = hip.MAX_DEC_HIDDEN + 1
    assert lib.mmn_workspace_bytes(C.byref(m), 4096) == 0


def test_plan_create_rejects_bad_arguments(lib):
    m = _model()
    plan = C.c_void_p()
    assert lib.mmn_plan_create(C.byref(m), 0, None, 0, None, C.byref(plan)) == -1
    assert lib.mmn_plan_create(C.byref(m), 64, 1, 1 << 30, 4096, C.byref(plan)) == -3    # misaligned workspace


def test_dropout_entry_points_reject_null_arguments(lib):
    """mmn_draw_dropout / mmn_dropout_adopt (ABI 105) validate before they touch a device."""
    b = hip.Batch()
    p = (C.c_float * hip.MAX_ENCODERS)()
    assert lib.mmn_dropout_adopt(None, C.byref(b), p, 16, 64, None) == -1
    assert lib.mmn_draw_dropout(None, C.byref(b), p, 1, 16, 64, None) == -1
    assert lib.mmn_dropout_floats(None, 8) == 0


def test_missing_library_fails_loudly(tmp_path):
    with pytest.raises(hip.MmnError):
        hip.load(str(tmp_path / "libmmn_hip.so"))


def test_repack_tasks_stay_inside_their_parameter_tensors(lib):
    """Round 6 (the GPU abort of GPUTEST_r05): the decoders' shared backward operand Wdec^T was built by reading EVERY decoder's
    output weight as [2 x S]; an MLPDecoder with hidden layers has [2 x last hidden width], so the repack read up to
    2 S - 2 width floats past that tensor - past the end of the flat parameter buffer when the decoder is the model's last.
    build_layout now checks every repack task against its tensor on the host (a task that would leave it yields no plan):
    models with narrow-hidden MLPDecoders in last place must plan."""
    for S, hidden in ((100, (8,)), (128, (32, 32)), (16, (12, 7, 9)), (4, (5,))):
        m = hip.Model()
        m.state_size, m.n_encoders, m.n_decoders = S, 2, 2
        for e in range(2):
            me = m.enc[e]
            me.n_features, me.n_layers, me.activation, me.kind = 6, 2, hip.ACT_RELU, hip.ENC_MIMIC
            me.layer[0].in_dim, me.layer[0].out_dim = 6 + S, 8
            me.layer[1].in_dim, me.layer[1].out_dim = 8, S
        m.dec[0].n_hidden = 0                                   # a ClassDecoder first, the MLPDecoder LAST in the flat buffer
        md = m.dec[1]
        md.n_hidden, md.hidden_activation = len(hidden), hip.ACT_RELU
        a = S
        for l, h in enumerate(hidden):
            md.hidden[l].in_dim, md.hidden[l].out_dim = a, h
            a = h
        assert lib.mmn_workspace_bytes(C.byref(m), 64) > 0, (S, hidden)
