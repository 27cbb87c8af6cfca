"""Static check of the hand-placed waits of lr_gload16_async (csrc/lr_scan.h) in compiler output (hipcc -S).

The scan loops issue their group loads from inline asm (saddr-form global_load_dwordx4) and wait for them by hand: the
compiler believes the destination registers hold the value as soon as the asm statement has run.  So on every path
from such a load, no instruction may touch its destination registers until a wait has made it land.  Vector-memory
loads return in order, and `s_waitcnt vmcnt(N)` returns once at most N vector-memory operations are outstanding: the
tracked load has landed at a wait with N <= (number of vector-memory instructions issued AFTER it on this path).  A
`vmcnt(1)` right behind the load (the hand-placed one that lets the next group's registers stay in flight, or one the
compiler inserts for its own loads) therefore does NOT end the tracking - treating any vmcnt as a full wait would pass a
later read of registers still in flight.
The walk is a depth-first search over (instruction, loads issued since) states along fall-through and branch edges.
"""
import re
import sys

# Instructions that increment vmcnt on gfx9 AND complete in issue order with the tracked global load (loads, and stores /
# atomics without return, of the global / buffer / scratch paths).  FLAT operations also use vmcnt but may be served by
# LDS and then complete out of order with global ones (which is why LLVM forces vmcnt(0) while one is pending): they are
# NOT counted as "behind" the tracked load.  That is the safe side: with `behind` in-order operations issued after the
# load, a wait that leaves at most N <= behind operations outstanding cannot leave the load outstanding, whatever else
# (a flat operation) is in the counter as well.
_VMEM = re.compile(r"^(global_|buffer_|scratch_|tbuffer_|image_)")
_LOAD = re.compile(r"\s+global_load_dwordx4 v\[(\d+):(\d+)\], v\d+, s\[\d+:\d+\]")


def _vmcnt(s):
    """N of an s_waitcnt that carries a vmcnt field, None if it has none (or is no s_waitcnt)."""
    if not s.startswith("s_waitcnt"):
        return None
    m = re.search(r"vmcnt\((\d+)\)", s)
    if m:
        return int(m.group(1))
    m = re.match(r"s_waitcnt\s+(0x[0-9a-fA-F]+|\d+)\s*$", s)       # raw immediate: vmcnt = bits 3:0 and 15:14
    if m:
        v = int(m.group(1), 0)
        return (v & 0xf) | ((v >> 14) & 0x3) << 4
    return None


def check(paths, verbose=True):
    """-> (number of hand-placed / saddr-form loads seen, number of violations)"""
    bad = 0
    total = 0
    for path in paths:
        bad_p, n_p = _check_one(path, verbose)
        bad += bad_p
        total += n_p
    return total, bad


def _check_one(path, verbose):
    bad = 0
    lines = [l.rstrip() for l in open(path)]
    labels = {}
    for i, l in enumerate(lines):
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            labels[m.group(1)] = i
    n_loads = 0
    for i, l in enumerate(lines):
        m = _LOAD.match(l)
        if not m:
            continue
        n_loads += 1
        lo, hi = int(m.group(1)), int(m.group(2))
        dst = set(range(lo, hi + 1))
        same = re.compile(r"global_load_dwordx4 v\[%d:%d\]," % (lo, hi))
        # state = (line, vector-memory instructions issued after the tracked load, capped); a line is revisited only
        # with FEWER loads behind it (fewer loads = harder to have landed = the stronger check)
        best = {}
        stack = [(i + 1, 0)]
        violated = False
        while stack and not violated:
            j, behind = stack.pop()
            while j < len(lines):
                if best.get(j, 1 << 30) <= behind:
                    break
                best[j] = behind
                s = lines[j].strip()
                if not s or (s.startswith((";", ".")) and not s.startswith(".LBB")) or s.endswith(":"):
                    j += 1
                    continue
                n = _vmcnt(s)
                if n is not None and n <= behind:
                    break                                   # the tracked load has landed on this path
                if s.startswith(("s_endpgm", "s_setpc")):
                    break
                regs = set()
                for a, b in re.findall(r"v\[(\d+):(\d+)\]", s):
                    regs.update(range(int(a), int(b) + 1))
                regs.update(int(r) for r in re.findall(r"\bv(\d+)\b", s))
                if same.match(s):
                    # the same asm load issued again into the same registers (the next trip): not a read; the tracked
                    # load is superseded - the new one is tracked from its own line
                    break
                if regs & dst:
                    print("%s:%d: %s   <- touches v[%d:%d] of the load at line %d before a wait that covers it (%d "
                          "vector-memory instructions behind it)" % (path, j + 1, s, lo, hi, i + 1, behind))
                    bad += 1
                    violated = True
                    break
                if _VMEM.match(s):
                    behind = min(behind + 1, 64)
                mb = re.match(r"(s_cbranch_\w+|s_branch)\s+(\.LBB\d+_\d+)", s)
                if mb:
                    if mb.group(2) in labels:
                        stack.append((labels[mb.group(2)], behind))
                    if mb.group(1) == "s_branch":
                        break
                j += 1
    if verbose:
        print(path, "async loads:", n_loads)
    return bad, n_loads


if __name__ == "__main__":
    n, bad = check(sys.argv[1:])
    print("violations:", bad)
    sys.exit(1 if bad else 0)
