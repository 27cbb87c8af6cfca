"""Static check of the hand-placed waits of lr_gload16_async (csrc/lr_scan.h) in compiler output (hipcc -S): on every
path from an saddr-form global_load_dwordx4 to the next s_waitcnt vmcnt, no instruction may touch the load's destination
registers (the compiler believes the value is there as soon as the asm statement has run)."""
import re, sys


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
    if True:
        lines = [l.rstrip() for l in open(path)]
        labels = {}
        for i, l in enumerate(lines):
            m = re.match(r"^(\.LBB\d+_\d+):", l)
            if m:
                labels[m.group(1)] = i
        n_loads = 0
        for i, l in enumerate(lines):
            m = re.match(r"\s+global_load_dwordx4 v\[(\d+):(\d+)\], v\d+, s\[\d+:\d+\]", l)
            if not m:
                continue
            n_loads += 1
            dst = set(range(int(m.group(1)), int(m.group(2)) + 1))
            seen, stack = set(), [i + 1]
            while stack:
                j = stack.pop()
                while j < len(lines) and j not in seen:
                    seen.add(j)
                    s = lines[j].strip()
                    if not s or s.startswith((";", ".")) and not s.startswith(".LBB") or s.endswith(":"):
                        j += 1
                        continue
                    if "s_waitcnt" in s and "vmcnt" in s:
                        break
                    if s.startswith(("s_endpgm", "s_setpc")):
                        break
                    regs = set()
                    for a, b in re.findall(r"v\[(\d+):(\d+)\]", s):
                        regs.update(range(int(a), int(b) + 1))
                    regs.update(int(r) for r in re.findall(r"\bv(\d+)\b", s))
                    # the same asm load issued again into the same registers (next trip) is not a read
                    if regs & dst and not re.match(r"global_load_dwordx4 v\[%d:%d\]," % (min(dst), max(dst)), s):
                        print("%s:%d: %s   <- touches v[%d:%d] of the load at line %d before a vmcnt wait" % (path, j + 1, s, min(dst), max(dst), i + 1))
                        bad += 1
                        break
                    mb = re.match(r"(s_cbranch_\w+|s_branch)\s+(\.LBB\d+_\d+)", s)
                    if mb:
                        if mb.group(2) in labels:
                            stack.append(labels[mb.group(2)])
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
