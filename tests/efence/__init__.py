"""GPU electric fence for the test suite (test infrastructure; the product never imports this).

`install()` swaps torch's caching allocator for tests/efence/libefence.so (efence_alloc.cpp): every device tensor
becomes its own mapping that ends flush against unmapped address space, so that a kernel reading or writing past the
end of any tensor faults deterministically.  Must run before the process makes its first device allocation.
`MMN_EFENCE=1 python -m pytest tests -m gpu` runs the whole suite that way (tests/conftest.py); tools/fault_hunt.py
runs the generic tier's random sweeps seed by seed in child processes under it."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "efence_alloc.cpp")
LIB = os.path.join(HERE, "libefence.so")


def build(force: bool = False) -> str:
    if force or not os.path.exists(LIB) or os.path.getmtime(LIB) < os.path.getmtime(SRC):
        hipcc = "/opt/rocm/bin/hipcc"
        subprocess.run([hipcc, "-O2", "-fPIC", "-shared", SRC, "-o", LIB], check=True)
    return LIB


_installed = False


def install() -> None:
    global _installed
    if _installed:
        return
    import torch
    if not os.path.exists(LIB):
        build()
    alloc = torch.cuda.memory.CUDAPluggableAllocator(LIB, "efence_malloc", "efence_free")
    torch.cuda.memory.change_current_allocator(alloc)
    # a free waits for the device before it unmaps - which must not happen inside a graph capture (the garbage collector may
    # run there): frees that arrive between __enter__ and __exit__ of torch.cuda.graph are parked by the library
    import ctypes
    lib = ctypes.CDLL(LIB)
    graph = torch.cuda.graph
    enter, leave = graph.__enter__, graph.__exit__

    def _enter(self):
        lib.efence_capture(1)
        return enter(self)

    def _exit(self, *exc):
        try:
            return leave(self, *exc)
        finally:
            lib.efence_capture(0)
    graph.__enter__, graph.__exit__ = _enter, _exit
    _installed = True
