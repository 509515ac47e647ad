"""The randomised parity cases of tests/debug/fuzz_gpu.py as functions: random batches of random shapes through both C ABIs and the
adapter against the CPU restatements.  Shared by the open-ended debug script (minutes, any seed) and tests/test_gpu_fuzz.py (a bounded,
fixed-seed slice that the -m gpu run executes).  Every function raises AssertionError with the offending shapes on a mismatch."""
import numpy as np

import oracle  # checker
from rovaca_amd import hostapi, swapi, synth
from rovaca_amd.batch import Batch

ACGT = np.frombuffer(b"ACGTN", np.uint8)


def pairhmm_iteration(eng, rng, it, long_bias=False, long_pair_cap=None, f64_first=False):
    """One random PairHMM batch (every kernel family: single-pass, ragged, striped, band; general / Rovaca-case / one gap-open), through
    the one-shot call or -- every third iteration -- upload + two sweeps + download.  Returns the number of pairs checked.
    f64_first: the caller has set PHMM_F64_FIRST=1 -- long pairs are swept fp64 first, and the fp32 sum of a pair the library proves
    underflowed may be reported as 0 instead of its (discarded) fp32 value; everything else must still be the oracle's bits."""
    acgt = ACGT
    mode = int(rng.choice([2, 3, 3, 4])) if long_bias else int(rng.integers(0, 4))
    rmax = [40, 260, 700, 2600, 7000][mode]
    hmax = [80, 520, 1500, 5000, 9000][mode]
    n_reads, n_haps = int(rng.integers(1, [40, 30, 10, 4, 3][mode])), int(rng.integers(1, [12, 10, 5, 3, 3][mode]))
    haps = [acgt[rng.integers(0, 5 if rng.random() < 0.2 else 4, int(rng.integers(1, hmax)))] for _ in range(n_haps)]
    reads = []
    general = rng.random() < 0.3
    uniform_gop = None if general or rng.random() < 0.5 else int(rng.choice([45, 40, 33, 20]))  # one gap-open in the batch
    for _ in range(n_reads):
        R = int(rng.integers(1, rmax))
        hap = haps[int(rng.integers(0, n_haps))]
        if len(hap) >= R and rng.random() < 0.8:
            a = int(rng.integers(0, len(hap) - R + 1))
            seq = hap[a:a + R].copy()
            seq[rng.random(R) < rng.choice([0.0, 0.01, 0.1])] = ord("G")
        else:
            seq = acgt[rng.integers(0, 4, R)]
        q = rng.integers(2, 60, R).astype(np.uint8)
        if general:
            gi, gd, gc = (rng.integers(5, 50, R).astype(np.uint8) for _ in range(3))
        elif uniform_gop is not None:
            gi = gd = np.full(R, uniform_gop, np.uint8)
            gc = np.full(R, 10, np.uint8)
        else:
            gi = gd = rng.choice(np.array([45, 40, 35], np.uint8), R)
            gc = np.full(R, 10, np.uint8)
        reads.append((seq, q, gi, gd, gc))
    big = mode < 2 and it % 7 == 0  # now and then a launch big enough for the four-pairs-per-wave shapes
    n_pairs = int(rng.integers(2000, 7000) if big else rng.integers(1, 200))
    if long_pair_cap is not None and mode >= 2:
        n_pairs = min(n_pairs, long_pair_cap)  # (the CPU restatement needs a second per 1e8 cells)
    pairs = [(int(rng.integers(0, n_reads)), int(rng.integers(0, n_haps))) for _ in range(n_pairs)]
    b = Batch.from_lists(reads, haps, pairs)
    if it % 3 == 0:  # resident path: upload once, sweep twice, download
        rb = eng.upload(b)
        rb.sweep()
        rb.sweep()
        raw, l10, st = rb.download()
        rb.release()
    else:
        raw, l10, st = eng.compute(b)
    oraw, ol10, oresc = oracle.batch(b, n_threads=8)
    same_raw = raw.view(np.uint32) == oraw.view(np.uint32)
    if f64_first:
        same_raw |= oresc & (raw == 0)
    if not same_raw.all():
        bad = np.nonzero(~same_raw)[0]
        raise AssertionError(f"PairHMM fp32 mismatch, iteration {it}: pairs {bad[:5]}, shapes "
                             f"{[(len(reads[pairs[k][0]][0]), len(haps[pairs[k][1]])) for k in bad[:5]]}, general {general}, mode {mode}")
    fin = np.isfinite(ol10)
    ok = (np.array_equal(np.isfinite(l10), fin) and np.array_equal(l10[~oresc], ol10[~oresc]) and
          (not (oresc & fin).any() or np.abs(l10[oresc & fin] - ol10[oresc & fin]).max() < 1e-9))
    if not ok:
        bad = [k for k in range(len(pairs)) if not ((l10[k] == ol10[k]) or (np.isnan(l10[k]) and np.isnan(ol10[k])))]
        raise AssertionError(f"PairHMM log10 mismatch, iteration {it}, general {general}, mode {mode}: " + "; ".join(
            f"pair {k} R {len(reads[pairs[k][0]][0])} H {len(haps[pairs[k][1]])} rescued {bool(oresc[k])} gpu {l10[k]} oracle {ol10[k]}" for k in bad[:5]))
    assert int(st.n_rescued) == int(oresc.sum())
    return len(pairs)


def smith_waterman_iteration(al, rng, it):
    """One random Smith-Waterman batch (every lane shape, lengths up to the 2 047 limit, both parameter sets).  Returns the alignments checked."""
    acgt = ACGT
    n_ref = int(rng.integers(1, 6))
    lim = int(rng.choice([40, 300, 700, 2047]))
    refs = [acgt[rng.integers(0, 4, int(rng.integers(1, lim + 1)))] for _ in range(n_ref)]
    seqs, sp = [], []
    big_sw = it % 11 == 0 and lim <= 300
    for _ in range(int(rng.integers(2100, 3000) if big_sw else rng.integers(1, 40))):
        ref = refs[int(rng.integers(0, n_ref))]
        n = int(rng.integers(1, min(len(ref), lim) + 1))
        if rng.random() < 0.8:
            a = int(rng.integers(0, len(ref) - n + 1))
            s = ref[a:a + n].copy()
            s[rng.random(n) < rng.choice([0.0, 0.02, 0.2])] = ord("T")
            if rng.random() < 0.4 and n > 2:
                k = int(rng.integers(1, n - 1))
                s = np.concatenate([s[:k], acgt[rng.integers(0, 4, int(rng.integers(1, 12)))], s[k:]])[:2047]
        else:
            s = acgt[rng.integers(0, 4, int(rng.integers(1, lim + 1)))]
        seqs.append(s)
        sp.append((int(rng.integers(0, n_ref)), len(seqs) - 1))
    params = swapi.PARAMS_ASSEMBLE if rng.random() < 0.5 else swapi.PARAMS_GENOTYPE
    out, cig, _ = al.align(refs, seqs, sp, params)
    for p, (r, s) in enumerate(sp):
        o = oracle.sw_align(refs[r], seqs[s], params)
        got = (int(out[p]["score"]), int(out[p]["max_i"]), int(out[p]["max_j"]), int(out[p]["alignment_offset"]))
        if got != (o["score"], o["max_i"], o["max_j"], o["alignment_offset"]) or not np.array_equal(swapi.cigar_of(out[p], cig), o["cigar"]):
            raise AssertionError(f"Smith-Waterman mismatch, iteration {it}, pair {p}: lengths {len(refs[r])} x {len(seqs[s])}, got {got}, want "
                                 f"{(o['score'], o['max_i'], o['max_j'], o['alignment_offset'])}, cigars "
                                 f"{oracle.cigar_string(swapi.cigar_of(out[p], cig))} / {oracle.cigar_string(o['cigar'])}")
    return len(sp)


def adapter_iteration(rng):
    """One random region through rovaca::call_pairhmm (hostapi.init_pairhmm_ptr_gpu must have run) against the oracle's call_region."""
    acgt = ACGT
    s = int(rng.integers(0, 1 << 30))
    rl = (int(rng.integers(5, 60)), int(rng.integers(60, 260)))
    hl = int(rng.integers(rl[1], rl[1] + 300))
    b = synth.make_batch(seed=s, n_regions=1, reads=int(rng.integers(1, 60)), haps=int(rng.integers(1, 12)), read_len=rl, hap_len=hl,
                         sub_rate=float(rng.choice([0.0, 0.01, 0.05])))
    n = b.read_bases.shape[0]
    qual = rng.choice(np.array([2, 11, 25, 37, 41], np.uint8), n, p=[0.02, 0.05, 0.15, 0.7, 0.08])
    mq = rng.choice(np.array([60, 60, 60, 29, 12], np.uint8), b.n_reads)
    bases = b.read_bases.copy()
    for r in rng.choice(b.n_reads, min(b.n_reads, int(rng.integers(0, 4))), replace=False):
        o, L = int(b.read_off[r]), int(b.read_len[r])
        bases[o:o + L] = acgt[rng.integers(0, 4, L)]
    model = str(rng.choice(["NONE", "CONSERVATIVE", "HOSTILE", "AGGRESSIVE"]))
    minq = int(rng.choice([6, 18, 30]))
    got, surv = hostapi.call_region(bases, qual, b.read_off, b.read_len, mq, b.hap_bases, b.hap_off, b.hap_len, min_quality_threshold=minq,
                                    pcr_model=model)
    want = np.full(b.n_haps * b.n_reads, np.nan)
    wsurv = np.zeros(b.n_reads, np.uint32)
    k = oracle.lib().phmm_oracle_call_region(bases, qual, b.read_off, b.read_len, mq, b.n_reads, b.hap_bases, b.hap_off, b.hap_len, b.n_haps,
                                             minq, hostapi.PCR[model], want, wsurv)
    want = want.reshape(b.n_haps, b.n_reads)
    if not (len(surv) == k and np.array_equal(surv, wsurv[:k]) and np.array_equal(got[:, :k], want[:, :k]) and np.all(np.isnan(got[:, k:]))):
        raise AssertionError(f"region mismatch: generator seed {s}, model {model}, min_q {minq}, {b.n_reads} reads x {b.n_haps} haplotypes, "
                             f"read lengths {rl}, haplotype length {hl}")
