"""Which tensor stands in front of a faulting address?  `python tests/efence/whose.py <EFENCE_LOG file> <hex address>`:
the runtime's fault message names a page address; the allocation log (EFENCE_LOG) has every allocation's pointer, size
and the first unmapped address behind it."""
import sys


def main(path: str, addr_s: str) -> None:
    addr = int(addr_s, 16)
    live = {}
    for line in open(path):
        f = line.split()
        if f[0] == "A":
            live[int(f[1], 16)] = (int(f[2]), int(f[4], 16))
        elif f[0] == "F":
            live.pop(int(f[1], 16), None)
    best = None
    for ptr, (size, end) in live.items():
        if end <= addr < end + (1 << 21) and (best is None or end > best[2]):
            best = (ptr, size, end)
        if ptr - (1 << 21) <= addr < ptr and best is None:
            print(f"address {addr:#x} lies {ptr - addr} bytes IN FRONT of allocation {ptr:#x} ({size} bytes)")
    if best:
        print(f"address {addr:#x} is {addr - best[2]} bytes past the fence of allocation {best[0]:#x} ({best[1]} bytes)")
    else:
        print("no live allocation ends in front of that address")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
