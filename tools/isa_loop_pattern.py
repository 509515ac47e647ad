"""Print the op-class pattern of the steady loop (tuning aid). classes: m mul, f fma/fmac, a add_co, c cndmask, D dpp, + add, s salu, n s_nop"""
import re, sys
text = open(sys.argv[1]).read()
for K in [int(x) for x in sys.argv[2].split(",")]:
    m = re.search(r"^_ZN4phmm23pairhmm_systolic_kernelIfLi%dELb1EEEvNS_10KernelArgsE:.*?s_endpgm" % K, text, re.S | re.M)
    lines = m.group(0).split("\n")
    hdr = [i for i, l in enumerate(lines) if "Loop Header" in l]
    best = None
    for h in hdr:
        lab = lines[h].split(":")[0].strip()
        for j in range(h + 1, len(lines)):
            if re.search(r"s_c?branch\w* " + re.escape(lab) + r"\b", lines[j]):
                body = [l.strip() for l in lines[h:j + 1] if l.strip() and not l.strip().startswith(";")]
                if best is None or len(body) > len(best):
                    best = body
                break
    def cls(l):
        o = l.split()[0]
        if "dpp" in o: return "D"
        if o.startswith("v_cndmask"): return "c"
        if o.startswith("v_add_co"): return "a"
        if o.startswith("v_mul"): return "m"
        if o.startswith("v_fma"): return "f"
        if o.startswith("s_nop"): return "n" + l.split()[1]
        if o.startswith("s_"): return "s"
        if o.startswith("v_add_f32"): return "+"
        if o.startswith("scratch"): return "S"
        return "?"
    print(K, "".join(cls(l) for l in best[1:]))
