"""Multi-GPU partitioning of the decode path.

Stripes and columns are independent units of work (stripe.rs:154-165 builds an independent byte
map per (column, kind); every stripe has its own footer, dictionary and encodings), so the path
shards with no data-path collective: one process per GPU decodes its share.  The only exchange is
a tiny all-gather of per-rank row counts (and string byte totals) so that every rank knows the
global row offsets of its batches -- RCCL over xGMI when the backend is "nccl", gloo in the CPU
tests.
"""


def stripe_shard(n_stripes, rank, world):
    """Round-robin stripe -> rank map (config C5): stripe i goes to rank i % world."""
    return [i for i in range(n_stripes) if i % world == rank]


def column_shard(costs, world):
    """Greedy longest-processing-time balance of columns by their Arrow output bytes (config C4:
    l_comment ~31 B/row vs 16 B decimals vs 4 B dates).  Returns a list of column-index lists."""
    order = sorted(range(len(costs)), key=lambda i: -costs[i])
    loads = [0] * world
    out = [[] for _ in range(world)]
    for i in order:
        r = min(range(world), key=lambda k: (loads[k], k))
        out[r].append(i)
        loads[r] += costs[i]
    for cols in out:
        cols.sort()
    return out


def gather_counts(values, dist=None, device=None):
    """All-gather a short list of int64 counters (rows decoded, value bytes, error word) over the
    process group; returns a list with one list per rank.  With no process group: [values]."""
    if dist is None or not dist.is_initialized():
        return [list(values)]
    import torch
    t = torch.tensor(list(values), dtype=torch.int64, device=device or "cpu")
    out = [torch.zeros_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(out, t)
    return [[int(x) for x in o.tolist()] for o in out]


def global_row_offsets(all_counts, slot=0):
    """Exclusive prefix of the per-rank row counts: first global row of every rank."""
    offs, acc = [], 0
    for c in all_counts:
        offs.append(acc)
        acc += c[slot]
    return offs, acc
