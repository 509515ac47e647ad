import os, sys
import numpy as np
sys.path.insert(0, os.getcwd())
os.environ["PHMM_ENABLE_TEST_HOOKS"] = "1"; os.environ["PHMM_F64_FIRST"] = "0"
os.environ["PHMM_FORCE_LANES"] = "16"; os.environ["PHMM_NO_UNIFORM_GOP"] = "1"
from rovaca_amd import capi, synth
from rovaca_amd.batch import Batch
eng = capi.Engine(0)
K = 16
parts = [synth.make_batch(seed=K, n_regions=6, reads=12, haps=5, read_len=16 * K - 3, hap_len=(40 + 10 * K, 60 + 20 * K)),
         synth.make_batch(seed=K + 50, n_regions=6, reads=12, haps=5, read_len=16 * (K - 1) - 2, hap_len=(40 + 10 * K, 60 + 20 * K))]
b = Batch.concat(parts)
for trial in range(3):
    raw, l10, st = eng.compute(b)
    os.environ["PHMM_NO_TAB"] = "1"
    raw0, _, st0 = eng.compute(b)
    del os.environ["PHMM_NO_TAB"]
    bad = raw.view(np.uint32) != raw0.view(np.uint32)
    print("trial", trial, "bad", int(bad.sum()))
order, bins = capi.plan_bins(b)
print("bins", bins)
rl = b.read_len[b.pair_read]; hl = b.hap_len[b.pair_hap]
for rows, first, count in bins:
    seg = order[first:first + count]
    nb = 0
    for w in range(0, count, 16):
        g = seg[w:w + 16]
        rd = b.pair_read[g]
        ntab = 1 + int((rd[1:] != rd[:-1]).sum())
        bw = bad[g]
        if bw.any():
            nb += 1
            print(" bin rows", rows, "wg", w // 16, "tables", ntab, "bad groups", np.nonzero(bw)[0].tolist(), "R", rl[g].tolist(), "H", hl[g].tolist())
    print("bin rows", rows, "count", count, "wgs with bad", nb)
for s in (1, 2, 3, 4, 6, 8, 16):
    os.environ["PHMM_TAB_SLOTS"] = str(s)
    raw3, _, _ = eng.compute(b)
    print("TAB_SLOTS", s, "bad", int((raw3.view(np.uint32) != raw0.view(np.uint32)).sum()))
del os.environ["PHMM_TAB_SLOTS"]
raw, l10, st = eng.compute(b)
bad = raw.view(np.uint32) != raw0.view(np.uint32)
reads_bad = sorted(set(int(x) for x in b.pair_read[bad]))
print("bad reads", reads_bad)
for r in reads_bad[:6]:
    sel = np.nonzero(b.pair_read == r)[0]
    bases = b.read(r)[0]
    print(" read", r, "R", int(b.read_len[r]), "pairs", sel.size, "bad", int(bad[sel].sum()), "hasN", bool((bases == ord("N")).any()), "roff", int(b.read_off[r]), "roff%4", int(b.read_off[r]) % 4, "haps", [int(h) for h in b.pair_hap[sel]], "hoff%4", [int(b.hap_off[h]) % 4 for h in b.pair_hap[sel]])
good = [r for r in range(b.n_reads) if r not in reads_bad and b.read_len[r] in (209, 220)]
for r in good[:6]:
    sel = np.nonzero(b.pair_read == r)[0]
    bases = b.read(r)[0]
    print(" good read", r, "R", int(b.read_len[r]), "hasN", bool((bases == ord("N")).any()), "roff%4", int(b.read_off[r]) % 4, "hoff%4", [int(b.hap_off[h]) % 4 for h in b.pair_hap[sel]])
# the bad reads alone
sub = b.subset(np.nonzero(np.isin(b.pair_read, reads_bad))[0])
r1, _, _ = eng.compute(sub)
os.environ["PHMM_NO_TAB"] = "1"
r0, _, _ = eng.compute(sub)
del os.environ["PHMM_NO_TAB"]
print("bad reads alone: pairs", sub.n_pairs, "bad", int((r1.view(np.uint32) != r0.view(np.uint32)).sum()))
