"""Renumber the VGPRs of chosen kernels in hipcc's device assembly (tuning aid + build step).

Why: on gfx950 the run time of the sweep loop depends on WHICH physical VGPRs hold the operands, at identical instruction
streams (measured: renaming one shared constant from v64 to v65 costs 3 %; random renumberings of the steady loop spread
over 7 %).  hipcc offers no control over the assignment, so the build can apply a fixed permutation to the registers of a
kernel after code generation: a bijection on the single 32-bit VGPRs of the function, leaving alone v0 (work-item id on
entry), every register that appears in a tuple v[a:b] anywhere in the function, and all metadata.  Semantics are unchanged
by construction (pure renaming inside one function; the parity tests still run on the result).

usage: regperm.py in.s out.s spec.json        spec = {"<mangled kernel name>": {"seed": n} | {"map": {"17": 40, ...}}, ...}
       regperm.py --free in.s <kernel>        list the registers a permutation may touch
"""
import json
import random
import re
import sys

TOK = re.compile(r"\bv(\d+)\b")
TUP = re.compile(r"\bv\[(\d+):(\d+)\]")


def function_span(lines, name):
    start = next(i for i, l in enumerate(lines) if l.startswith(name + ":"))
    end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
    return start + 1, end


def split_comment(line):
    i = line.find(";")
    return (line, "") if i < 0 else (line[:i], line[i:])


def free_registers(lines, span):
    used, fixed = set(), {0}
    for l in lines[span[0]:span[1]]:
        code, _ = split_comment(l)
        for m in TUP.finditer(code):
            fixed.update(range(int(m.group(1)), int(m.group(2)) + 1))
        used.update(int(m.group(1)) for m in TOK.finditer(code))
    return sorted(used - fixed), sorted(fixed & (used | fixed))


def apply(lines, span, mapping):
    out = list(lines)
    for i in range(span[0], span[1]):
        code, com = split_comment(lines[i])
        if code.lstrip().startswith("."):
            continue
        out[i] = TOK.sub(lambda m: "v%d" % mapping.get(int(m.group(1)), int(m.group(1))), code) + com
    return out


def mapping_from(spec, free):
    if "map" in spec:
        mp = {int(k): int(v) for k, v in spec["map"].items()}
        assert set(mp) <= set(free) and sorted(mp.values()) == sorted(mp), "not a permutation of free registers"
        return mp
    rnd = random.Random(spec["seed"])
    tgt = list(free)
    rnd.shuffle(tgt)
    return dict(zip(free, tgt))


def main():
    if sys.argv[1] == "--free":
        lines = open(sys.argv[2]).read().split("\n")
        span = function_span(lines, sys.argv[3])
        free, fixed = free_registers(lines, span)
        print("free", len(free), free)
        print("fixed", fixed)
        return
    src, dst, specf = sys.argv[1:4]
    lines = open(src).read().split("\n")
    spec = json.load(open(specf))
    for name, sp in spec.items():
        span = function_span(lines, name)
        free, _ = free_registers(lines, span)
        lines = apply(lines, span, mapping_from(sp, free))
    open(dst, "w").write("\n".join(lines))


if __name__ == "__main__":
    main()
