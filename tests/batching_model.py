"""An independent statement of the CPU-side batching loops as prefix sums (numpy), used to cross-check oracle/batching_oracle.c.

The oracle restates the reference's loops literally (vectors that grow, a running batch set).  This file states WHAT those loops
compute in closed form -- every index is an exclusive prefix sum over bins or items -- which is also the formulation the device
kernels use.  Agreement of the two on seeded cases plus the hand-worked literals in tests/test_oracle_batching.py is the check."""
import numpy as np

NO = 0xFFFFFFFF


def cpu_bins(rows, row_kind, row_bin, row_input, unb_indexed, bat_indexed, no_indirect, initial):
    """initial = (wi[2], ip[2], bs[2], data).  Returns dict like oracle_lib.batch_cpu_bins (arrays absolute, zeros below initial)."""
    rows = np.asarray(rows, np.int64)
    kind, rbin, rin = np.asarray(row_kind)[rows], np.asarray(row_bin)[rows].astype(np.int64), np.asarray(row_input)[rows]
    wi_len, ip_len, bs_len, data = list(initial[0]), list(initial[1]), list(initial[2]), int(initial[3])
    wi = [[(0, 0)] * wi_len[c] for c in range(2)]
    md = [[(0, 0, 0, 0, 0)] * ip_len[c] for c in range(2)]
    bs = [[(0, 0)] * bs_len[c] for c in range(2)]
    unb, rec = [], []
    for b in range(len(unb_indexed)):
        sel = (kind == 2) & (rbin == b)
        n_all = int(sel.sum())
        if n_all == 0:
            continue
        c = int(unb_indexed[b])
        inputs = rin[sel]
        with_input = inputs[inputs != NO]
        k = len(with_input)
        out = data + np.arange(k)
        data += k
        if no_indirect:
            wi[c] += [(int(i), int(o)) for i, o in zip(with_input, out)]
            unb += [(b, int(o)) for o in out]
        else:
            ip = ip_len[c] + np.arange(k)
            md[c] += [(int(o), NO, 0, 0, 0) for o in out] + [(0, 0, 0, 0, 0)] * (n_all - k)   # allocate(len): the tail stays zero
            ip_len[c] += n_all
            wi[c] += [(int(i), int(p)) for i, p in zip(with_input, ip)]
            bs[c] += [(0, int(p)) for p in ip]
            unb += [(b, int(p)) for p in ip]
    for b in range(len(bat_indexed)):
        sel = (kind == 1) & (rbin == b)
        k = int(sel.sum())
        if k == 0:
            continue
        c = int(bat_indexed[b])
        out0 = data
        data += k
        if no_indirect:
            wi[c] += [(int(i), out0 + j) for j, i in enumerate(rin[sel])]
            rec.append((0x80000000 | b, c, 0, 0, k, NO, 1, out0))
        else:
            ip = ip_len[c]
            ip_len[c] += 1
            bsi = len(bs[c])
            md[c].append((out0, bsi, 0, 0, 0))
            bs[c].append((0, ip))
            wi[c] += [(int(i), ip) for i in rin[sel]]
            rec.append((0x80000000 | b, c, bsi, 0, k, ip, 1, out0))
    u32 = lambda a, w: np.array(a, np.uint32).reshape(-1, w)
    return dict(work_items=[u32(wi[c], 2) for c in range(2)], metadata=[u32(md[c], 5) for c in range(2)],
                batch_sets=[u32(bs[c], 2) for c in range(2)], unbatchable=u32(unb, 2), records=u32(rec, 8),
                totals=dict(work_item_len=[len(wi[0]), len(wi[1])], indirect_parameters_len=[len(md[0]), len(md[1])],
                            batch_set_len=[len(bs[0]), len(bs[1])], data_buffer_len=data))


def sorted_phase(items, automatic, no_indirect, initial):
    """Closed form of the sorted-phase walk: flags from adjacent items, everything else prefix sums over the items."""
    it = np.asarray(items, np.uint32).reshape(-1, 4)
    n = len(it)
    has_in = it[:, 0] != NO
    cls = (it[:, 3] & 1).astype(np.int64)
    meta = has_in & bool(automatic) & ((it[:, 3] & 2) != 0)
    prev_ok = np.zeros(n, bool)          # a batch set is alive when the previous item had an input index
    prev_ok[1:] = has_in[:-1]
    same_set = np.zeros(n, bool)
    same_bin = np.zeros(n, bool)
    same_set[1:] = meta[1:] & meta[:-1] & (it[1:, 1] == it[:-1, 1])
    same_bin[1:] = same_set[1:] & (it[1:, 2] == it[:-1, 2])
    ok = has_in & prev_ok & same_bin
    brk = has_in & prev_ok & same_set & ~same_bin & (not no_indirect)
    head = has_in & ~ok & ~brk
    out_index = int(initial[3]) + np.cumsum(has_in) - has_in            # exclusive count of items with input
    alloc = (head | brk) & (not no_indirect)
    ip_index = np.zeros(n, np.int64)
    for c in range(2):
        a = alloc & (cls == c)
        ip_index[a] = int(initial[1][c]) + (np.cumsum(a) - a)[a]
    head_of = np.maximum.accumulate(np.where(head, np.arange(n), -1))   # the item's batch set = latest head at or before it
    brk_incl = np.cumsum(brk)
    wi = [[(0, 0)] * int(initial[0][c]) for c in range(2)]
    md = [np.zeros((int(initial[1][c]) + int((alloc & (cls == c)).sum()), 5), np.uint32) for c in range(2)]
    bs = [[(0, 0)] * int(initial[2][c]) for c in range(2)]
    batches = []
    for i in range(n):
        if not has_in[i]:
            continue
        h = head_of[i]
        if alloc[i]:
            md[cls[i]][ip_index[i]] = (out_index[i], NO, 0, 0, 0)
        cur = ip_index[h] + (brk_incl[i] - brk_incl[h])               # range.end - 1
        wi[cls[i]].append((int(it[i, 0]), int(out_index[i]) if no_indirect else int(cur)))
        last = i + 1 == n or not has_in[i + 1] or head[i + 1]
        if last:
            if no_indirect:
                batches.append((h, out_index[h], out_index[i] + 1, NO, NO, cls[h]))
            else:
                batches.append((h, out_index[h], out_index[i] + 1, ip_index[h], cur + 1, cls[h]))
                bs[cls[h]].append((0, int(ip_index[h])))
    u32 = lambda a, w: np.array(a, np.uint32).reshape(-1, w)
    return dict(work_items=[u32(wi[c], 2) for c in range(2)], metadata=md, batch_sets=[u32(bs[c], 2) for c in range(2)],
                batches=u32(batches, 6),
                totals=dict(work_item_len=[len(wi[0]), len(wi[1])], indirect_parameters_len=[len(md[0]), len(md[1])],
                            batch_set_len=[len(bs[0]), len(bs[1])], data_buffer_len=int(initial[3]) + int(has_in.sum())))


def sorted_merge(items, automatic, first_index):
    it = np.asarray(items, np.uint32).reshape(-1, 4)
    n = len(it)
    has = it[:, 0] != NO
    meta = has & bool(automatic) & ((it[:, 3] & 2) != 0)
    index = first_index + np.cumsum(has) - has
    join = np.zeros(n, bool)
    join[1:] = meta[1:] & meta[:-1] & (it[1:, 1] == it[:-1, 1]) & (it[1:, 2] == it[:-1, 2])
    out = []
    for i in range(n):
        if join[i]:
            out[-1][2] = index[i] + 1
        elif has[i]:
            out.append([i, index[i], index[i] + 1, NO, NO, it[i, 3] & 1])
    return np.array(out, np.uint32).reshape(-1, 6), int(first_index + has.sum())
