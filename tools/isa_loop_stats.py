"""Count the instruction mix of the largest loop of each pairhmm_systolic_kernel<float,K,true> in a hipcc -S dump (tuning aid)."""
import re
import sys

text = open(sys.argv[1]).read()
ks = [int(x) for x in sys.argv[2].split(",")]
for K in ks:
    m = re.search(r"^_ZN4phmm23pairhmm_systolic_kernelIfLi%dELb1ELi16EEEvNS_10KernelArgsE:.*?s_endpgm" % K, text, re.S | re.M)
    lines = m.group(0).split("\n")
    hdr = [i for i, l in enumerate(lines) if "Loop Header" in l]
    best = None
    for h in hdr:
        lab = lines[h].split(":")[0].strip()
        for j in range(h + 1, len(lines)):
            if re.search(r"s_c?branch\w* " + re.escape(lab) + r"\b", lines[j]):
                body = [l for l in lines[h:j + 1] if l.strip() and not l.strip().startswith(";")]
                if best is None or len(body) > len(best):
                    best = body
                break
    b = best

    def cnt(pat):
        return sum(1 for l in b if re.search(pat, l))

    salu = cnt(r"^\s*s_")
    print("K=%d: loop %d instr: fp %d addco %d cnd %d dpp %d mov %d add %d salu %d scratch %d lds %d" % (
        K, len(b), cnt("v_fmac|v_mul_f32|v_fma_f32"), cnt("v_add_co"), cnt("v_cndmask"), cnt("_dpp"),
        cnt("v_mov_b32_e32|v_mov_b64"), cnt("v_add_f32"), salu, cnt("scratch_"), cnt("ds_")))
