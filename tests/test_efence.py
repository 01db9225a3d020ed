"""The GPU electric fence (tests/efence) and what it guards (round 6).

GPUTEST_r05 aborted in tests/test_mimic_family.py's random sweep with a GPU memory-access fault that three full-suite
runs of the builder had not shown: k_prepare's repack read an MLPDecoder's [2 x width] output weight as [2 x S] - past
the end of the flat parameter buffer when that decoder is the model's last - which faults only where the buffer ends its
allocator segment (multimodn_amd/csrc/mmn_host.inc, build_layout).  Under the fence every device
tensor ends flush against unmapped address space: such an overrun faults on every box, every time.

Each case runs in a CHILD process (a fault aborts the process that takes it; the allocator must be swapped before the first
device allocation):
  * the fence is live: a deliberate overrun - a NaN scan told that a batch has 4096 rows more than its tensor - is a fault;
  * the sweep seeds that faulted before the fix (and a sample of the others, both sweeps) run clean under it, flush to 16 and
    to 4 bytes, with host-staged and with device-resident inputs.
The whole `-m gpu` suite runs under it too (MMN_EFENCE=1 python -m pytest tests -m gpu; tools/fault_hunt.py per seed):
round 6: 481 passed after the fix, 540 at the end of the round."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

LIVE = r'''
import sys
sys.path.insert(0, %(repo)r); sys.path.insert(0, %(tests)r)
import efence; efence.install()
import ctypes as C
import numpy as np, torch
import multimodn_amd as mm
from helpers import build_torch_model
from oracle import multimodn_oracle as O
spec = O.ModelSpec(16, [O.EncoderSpec(8, (8,), O.ACT_RELU)], 1, 1.0, 0.3)
model = build_torch_model(spec, O.init_params(spec, 0), "cuda", mm)
eng = model._get_engine(8192)
x = torch.zeros(64, 8, device="cuda"); y = torch.zeros(64, 1, dtype=torch.int64, device="cuda")
b = eng.make_batch([x], y, [(0, 0)], device_nan_flags=True)
eng.nan_scan(b); torch.cuda.synchronize()
print("IN BOUNDS OK", flush=True)
b.batch = b.batch_global = 64 + 4096     # the scan now walks 4096 rows past the end of x
eng.nan_scan(b); torch.cuda.synchronize()
print("NO FAULT", flush=True)
'''


def _child(code=None, args=None, env=None, timeout=600):
    e = dict(os.environ)
    e.update({"MMN_EPOCH_KERNEL": "0"})
    e.update(env or {})
    cmd = [sys.executable, "-c", code] if code is not None else [sys.executable] + args
    return subprocess.run(cmd, env=e, capture_output=True, text=True, timeout=timeout)


def test_the_fence_is_live():
    r = _child(LIVE % {"repo": REPO, "tests": os.path.join(REPO, "tests")})
    assert "IN BOUNDS OK" in r.stdout, (r.returncode, r.stdout[-500:], r.stderr[-1500:])
    assert r.returncode != 0 and "NO FAULT" not in r.stdout, "a 4096-row overrun of a device tensor did not fault under the fence"
    assert "fault" in r.stderr.lower() or r.returncode < 0, r.stderr[-1500:]


# seeds 6, 26, 28, 32, 102, 105: faulted in k_prepare before the fix (an MLPDecoder with hidden layers in last place)
@pytest.mark.parametrize("align,device_inputs", [("16", False), ("4", False), ("4", True)])
def test_generic_tier_sweeps_run_clean_under_the_fence(align, device_inputs):
    seeds = "6,26,28,32,7,10,11,12,13,30,40,42,47,0,1,2,102,105,106,111,112,129,130,100"
    args = [os.path.join(REPO, "tools", "fault_hunt.py"), "--child", "--seeds", seeds] + (["--device-inputs"] if device_inputs else [])
    r = _child(args=args, env={"MMN_EFENCE": "1", "EFENCE_ALIGN": align})
    ok = [ln for ln in r.stdout.splitlines() if ln.startswith("SEED") and ln.endswith(" ok")]
    bad = [ln for ln in r.stdout.splitlines() if "mismatch" in ln]
    assert r.returncode == 0 and not bad and len(ok) == len(seeds.split(",")), (r.returncode, bad, r.stdout[-800:], r.stderr[-1500:])
