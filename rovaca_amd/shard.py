"""Pair-list sharding for multi-GPU runs: pairs are independent, so a batch is cut into one slice per
rank with no exchange step.  Slices are contiguous ranges of the length-binned order (rows-per-lane, haplotype length) and
carry equal numbers of DP cells, not equal numbers of pairs.  The only communication is the final gather of results by pair
id, done on the host (gloo) or skipped when every rank keeps its own results."""
import numpy as np

LANES_PER_PAIR = 16


def binned_order(batch):
    """Indices of the pair list sorted by (rows per lane K = ceil(R/16), haplotype length): the order the engine sweeps."""
    r = batch.read_len[batch.pair_read].astype(np.int64)
    h = batch.hap_len[batch.pair_hap].astype(np.int64)
    k = (r + LANES_PER_PAIR - 1) // LANES_PER_PAIR
    return np.lexsort((np.arange(batch.n_pairs), h, k))


def partition_equal_cells(batch, world):
    """List of `world` index arrays (into the pair list), contiguous in the binned order, near-equal in total cells."""
    order = binned_order(batch)
    r = batch.read_len[batch.pair_read[order]].astype(np.int64)
    h = batch.hap_len[batch.pair_hap[order]].astype(np.int64)
    cum = np.cumsum(r * h)
    total = int(cum[-1]) if len(cum) else 0
    cuts = [0]
    for g in range(1, world):
        cuts.append(int(np.searchsorted(cum, total * g / world, side="left")))
    cuts.append(len(order))
    cuts = np.maximum.accumulate(np.array(cuts))
    return [order[cuts[g]:cuts[g + 1]] for g in range(world)]


def compute_sharded(batch, compute_fn, rank, world, dist=None):
    """Every rank evaluates its slice with compute_fn(sub_batch) -> (raw_f32, log10); results are gathered by pair id.
    Returns (raw_f32, log10) for the whole batch on every rank when `dist` (an initialised torch.distributed) is given,
    else the rank-local (indices, raw, log10)."""
    idx = partition_equal_cells(batch, world)[rank]
    raw, l10 = compute_fn(batch.subset(idx))
    if dist is None:
        return idx, raw, l10
    parts = [None] * world
    dist.all_gather_object(parts, (idx, np.asarray(raw), np.asarray(l10)))
    out_raw = np.zeros(batch.n_pairs, np.float32)
    out_l10 = np.zeros(batch.n_pairs, np.float64)
    for i, r, l in parts:
        out_raw[i] = r
        out_l10[i] = l
    return out_raw, out_l10
