import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)
HERE = os.path.dirname(os.path.abspath(__file__))
if HERE not in sys.path:
    sys.path.insert(0, HERE)


# The one-launch epoch kernel (MultiModN._small_epoch) would take every small device-resident training loop of this suite,
# including the ones written to compare the step-by-step tiers with each other: it is switched on where it is the subject
# (tests/test_epoch_small.py, and the shipping-default legs of tests/test_hip_parity.py set MMN_EPOCH_KERNEL=1).
os.environ.setdefault("MMN_EPOCH_KERNEL", "0")

# MMN_EFENCE=1: the whole session under the GPU electric fence (tests/efence): every device tensor ends flush against
# unmapped address space, an out-of-bounds access of any kernel faults on the spot.  Has to happen before the first device
# allocation of the process.
if os.environ.get("MMN_EFENCE", "0") not in ("", "0"):
    import efence
    efence.install()


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


# ---------------------------------------------------------------------------------------------------------------------
# Order of the session: `pytest -x` stops at the first failure and a GPU memory fault takes the whole
# process with it, so what a round is judged by runs FIRST - the reference-golden and BASELINE-config tests of every
# SURVEY section 8 row, per-sample mode (BASELINE configs[4]) included - then the remaining deterministic tests, then the
# seeded random sweeps, and the multi-process tests (which share one GPU between up to eight processes) last.
# Within a rank the files keep their order and the tests their order inside the file.
# ---------------------------------------------------------------------------------------------------------------------
_RANK_FIRST = 0      # goldens + BASELINE configs
_RANK_REST = 1
_RANK_SWEEP = 2      # seeded random sweeps, long full-size curves
_RANK_MP = 3         # multi-process

_FIRST_FILES = ("test_oracle_golden.py", "test_abi.py", "test_host_logic.py")
_GOLDEN_WORDS = ("reference_golden", "reference_run", "at_their_stated_sizes", "full_size", "c5_", "matches_oracle",
                 "match_oracle", "reference_written_checkpoint", "integration_md")
_SWEEP_WORDS = ("random_", "sweep_ran", "loss_curves_at_full_size")
_MP_WORDS = ("rank", "peers", "bench_launch", "bench_with_eight", "exchange_inside", "oneshot_exchange")


def _rank(item) -> int:
    fname = os.path.basename(str(item.fspath))
    name = item.name
    if fname == "test_dp_gloo.py" or (fname == "test_step_protocol.py" and any(w in name for w in ("two_ranks", "ranks_on"))):
        return _RANK_MP
    if any(w in name for w in _SWEEP_WORDS):
        return _RANK_SWEEP
    if fname in _FIRST_FILES or fname == "test_per_sample.py" or any(w in name for w in _GOLDEN_WORDS):
        return _RANK_FIRST
    return _RANK_REST


def pytest_collection_modifyitems(session, config, items):
    order = {id(it): k for k, it in enumerate(items)}
    items.sort(key=lambda it: (_rank(it), order[id(it)]))
