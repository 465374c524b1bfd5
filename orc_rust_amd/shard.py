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


def unit_shard(stripe_rows, column_costs, world):
    """(stripe, column) units balanced over the ranks by Arrow output bytes, longest first (LPT).  This is the partition
    of config C4 at scale: a column shard alone leaves the rank that holds l_comment (18 % of the bytes) with more than 1/8
    of the work, a stripe shard alone ignores that columns differ; units of one column of one stripe balance both ways.
    Returns (units, loads): units[rank] = sorted [(stripe, column)], loads[rank] = its byte estimate."""
    items = [(rows * cost, s, c) for s, rows in enumerate(stripe_rows) for c, cost in enumerate(column_costs)]
    items.sort(key=lambda t: (-t[0], t[1], t[2]))
    loads = [0.0] * world
    units = [[] for _ in range(world)]
    for w, s, c in items:
        r = min(range(world), key=lambda k: (loads[k], k))
        units[r].append((s, c))
        loads[r] += w
    for u in units:
        u.sort()
    return units, loads


def stripe_row_offsets(stripe_rows):
    """First global row of every stripe (file order), whichever rank decodes it."""
    offs, acc = [], 0
    for n in stripe_rows:
        offs.append(acc)
        acc += n
    return offs, acc


def check_unit_coverage(all_units, n_stripes, n_columns):
    """Every (stripe, column) unit decoded exactly once over all ranks?  Raises AssertionError otherwise."""
    seen = {}
    for rank, units in enumerate(all_units):
        for u in units:
            assert u not in seen, "unit %s decoded by ranks %d and %d" % (u, seen[u], rank)
            seen[u] = rank
    assert len(seen) == n_stripes * n_columns, "units decoded: %d of %d" % (len(seen), n_stripes * n_columns)


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
