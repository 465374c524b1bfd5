"""Restatement, in plain Python, of the reference's row-selection stepping -- TEST INFRASTRUCTURE ONLY:
  * `RowSelection::from(Vec<RowSelector>)`        src/row_selection.rs:466-482
  * `RowSelection::split_off`                      src/row_selection.rs:278-314
  * `NaiveStripeDecoder::next_with_row_selection`  src/array_decoder/mod.rs:302-365
  * how `ArrowReader` hands each stripe its share  src/arrow_reader.rs:296-308
Selectors are (row_count, skip) pairs."""


def normalise(selectors):
    out = []
    for n, skip in selectors:
        if n == 0:
            continue
        if out and out[-1][1] == skip:
            out[-1] = (out[-1][0] + n, skip)
        else:
            out.append((n, skip))
    return out


def split_off(sel, row_count):
    """Returns (first row_count rows, rest) like split_off (which mutates self into the rest)."""
    total = 0
    idx = None
    for i, (n, _) in enumerate(sel):
        total += n
        if total > row_count:
            idx = i
            break
    if idx is None:
        return list(sel), []
    head, remaining = list(sel[:idx]), list(sel[idx:])
    overflow = total - row_count
    if remaining[0][0] != overflow:
        head.append((remaining[0][0] - overflow, remaining[0][1]))
    remaining[0] = (overflow, remaining[0][1])
    return head, remaining


def stripe_batches(sel, number_of_rows, batch_size):
    """Row ranges (start, len) of the RecordBatches the stripe decoder yields under `sel`."""
    out, index, si = [], 0, 0
    while si < len(sel):
        row_count, skip = sel[si]
        remaining = number_of_rows - index
        if skip:
            actual = min(row_count, remaining)
            if actual == 0:
                si += 1
                continue
            index += actual
            if actual >= row_count:
                si += 1
        else:
            actual = min(row_count, batch_size, remaining)
            if actual == 0:
                si += 1
                continue
            out.append((index, actual))
            index += actual
            if actual >= row_count:
                si += 1
    return out


def file_batches(selectors, stripe_rows, batch_size):
    """Per stripe: list of (start, len) within the stripe, or None = the stripe is read whole (arrow_reader.rs:296-308:
    a selection with no rows left no longer applies)."""
    sel = normalise(selectors)
    out = []
    for n in stripe_rows:
        if sum(x[0] for x in sel) > 0:
            mine, sel = split_off(sel, n)
            out.append(stripe_batches(mine, n, batch_size))
        else:
            out.append(None)
    return out
