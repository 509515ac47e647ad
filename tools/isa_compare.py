"""Compare two device-assembly dumps (hipcc --save-temps: *-gfx950.s) kernel by kernel: same instruction stream or not, labels and
comments aside.  The single-pass PairHMM kernels are sensitive to register numbering and schedule ("Register numbers
matter"), so a change that is not meant to touch them is checked with this.  usage: python tools/isa_compare.py old.s new.s [name-filter]"""
import re
import sys


def funcs(path):
    s = open(path).read()
    out = {}
    for m in re.finditer(r"^(_Z\w+):[^\n]*\n(.*?)^\.Lfunc_end\d+:", s, re.S | re.M):
        body = re.sub(r";.*", "", m.group(2))
        body = re.sub(r"\.LBB\d+_\d+", "L", body)
        body = re.sub(r"^\s*\.amdhsa_kernarg_size.*$", "", body, flags=re.M)  # (the argument struct may grow at its end)
        out[m.group(1)] = body
    return out


old, new = funcs(sys.argv[1]), funcs(sys.argv[2])
flt = sys.argv[3] if len(sys.argv) > 3 else ""
same = diff = 0
for k in sorted(new):
    if flt not in k:
        continue
    if k not in old:
        print("only in new:", k)
    elif new[k] == old[k]:
        same += 1
    else:
        diff += 1
        print("DIFFERENT:", k, f"({len(old[k].splitlines())} -> {len(new[k].splitlines())} lines)")
for k in sorted(old):
    if flt in k and k not in new:
        print("only in old:", k)
print(f"{same} kernels identical, {diff} different")
