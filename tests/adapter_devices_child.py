"""Child process of tests/test_adapter_devices.py: the drop-in adapter with N batchers (init_pairhmm_ptr_gpu(-1)) -- on a one-GPU box the
device-list hook puts all of them on device 0.  30 worker threads call rovaca::call_pairhmm on regions of mixed shape, quality thresholds
and PCR models; every region's matrix is compared with the oracle's restatement of the reference call.  Prints one JSON line."""
import json
import os
import sys
import threading

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import oracle  # noqa: E402  (checker)
from rovaca_amd import hostapi, synth  # noqa: E402


def raw_region(seed, reads, haps, read_len, hap_len, bad_reads=0):
    rng = np.random.default_rng(seed)
    b = synth.make_batch(seed=seed, n_regions=1, reads=reads, haps=haps, read_len=read_len, hap_len=hap_len)
    n = b.read_bases.shape[0]
    qual = rng.choice(np.array([2, 11, 25, 37, 41], np.uint8), n, p=[0.02, 0.05, 0.15, 0.7, 0.08])
    mq = rng.choice(np.array([60, 60, 60, 29, 12], np.uint8), b.n_reads)
    bases = b.read_bases.copy()
    for r in rng.choice(b.n_reads, bad_reads, replace=False):
        o, L = int(b.read_off[r]), int(b.read_len[r])
        bases[o:o + L] = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, L)]
    return bases, qual, b.read_off, b.read_len, mq, b.hap_bases, b.hap_off, b.hap_len


def main():
    n_workers, per_worker = int(sys.argv[1]), int(sys.argv[2])
    assert hostapi.init_pairhmm_ptr_gpu(-1)
    models = ["NONE", "CONSERVATIVE", "HOSTILE"]
    regions = []
    for i in range(24):
        regions.append((raw_region(900 + i, reads=8 + 7 * (i % 6), haps=2 + i % 7, read_len=(40, 151), hap_len=180 + 13 * i, bad_reads=i % 3),
                        (18, 10)[i % 2], models[i % 3]))
    want = []
    for args, min_q, model in regions:
        n_reads, n_haps = len(args[3]), len(args[7])
        w = np.full(n_haps * n_reads, np.nan)
        ws = np.zeros(n_reads, np.uint32)
        k = oracle.lib().phmm_oracle_call_region(args[0], args[1], args[2], args[3], args[4], n_reads, args[5], args[6], args[7], n_haps,
                                                 min_q, hostapi.PCR[model], w, ws)
        want.append((w.reshape(n_haps, n_reads), ws[:k].copy()))
    bad = []

    def work(t):
        for j in range(per_worker):
            i = (t * 7 + j * 5) % len(regions)
            args, min_q, model = regions[i]
            got, surv = hostapi.call_region(*args, min_quality_threshold=min_q, pcr_model=model)
            w, ws = want[i]
            k = len(ws)
            if not (len(surv) == k and np.array_equal(surv, ws) and np.array_equal(got[:, :k], w[:, :k]) and np.all(np.isnan(got[:, k:]))):
                bad.append((t, j, i))

    th = [threading.Thread(target=work, args=(t,)) for t in range(n_workers)]
    [t.start() for t in th]
    [t.join() for t in th]
    print(json.dumps({"batchers": hostapi.batcher_stats(), "regions": n_workers * per_worker, "mismatches": bad[:5], "n_mismatches": len(bad)}))


if __name__ == "__main__":
    main()
