"""The closed forms of the point-tile-major workgroup order (csrc/mlp_infer.hip, InferArgs.pm_period; psn_mlp_block_order), restated in
Python and checked exhaustively on small cases: every 64-row block of the (group, point) row set is visited exactly once, for group
sizes that are no multiple of 64 (blocks that straddle two groups) and for the padded launches of the HIP-graph path (real blocks
first, the dumped groups' tiles behind them, surplus workgroups = the all-padding blocks).  The bit-identity of the launch itself is
a GPU test (tests/test_graph_gpu.py::test_point_major_block_order_is_bit_identical_to_row_order)."""
import itertools

import pytest


def ragged_order(n_rows, P, G):
    """blockIdx -> block of the unpadded launch (mirrors the `else if (g.pm_period > 0)` branch); None = the workgroup leaves."""
    btot = (n_rows + 63) // 64
    per = (btot + 7) // 8
    grid = (btot + 7) // 8 * 8
    nfull, r = P // 64, P % 64
    out = []
    for b in range(grid):
        v = (b & 7) * per + (b >> 3)
        if v >= btot:
            out.append(None)
            continue
        if v < nfull * G:
            pt, l = divmod(v, G)
        else:
            pt, l = nfull, ((v - nfull * G) * 64) // r
        out.append(l * nfull + (l * r + 63) // 64 + pt)
    return out, btot


@pytest.mark.parametrize('P,G', [(64, 3), (100, 5), (29487, 104), (4099, 16), (65, 2), (127, 7), (128, 9), (1000, 7), (1024, 104), (3686, 104)])
def test_ragged_point_major_order_is_a_bijection(P, G):
    order, btot = ragged_order(P * G, P, G)
    seen = [b for b in order if b is not None]
    assert len(seen) == btot and sorted(seen) == list(range(btot))
    # every block lies in the group the formula assigned it to: its FIRST row belongs to group l
    nfull, r = P // 64, P % 64
    for b in seen[:2000]:
        first_row = 64 * b
        l = first_row // P
        assert (l * nfull + (l * r + 63) // 64) <= b
    # an XCD (workgroups x, x + 8, ...) walks a contiguous eighth of the point-major order: consecutive workgroups of one XCD visit
    # the same point tile under consecutive groups
    per = (btot + 7) // 8
    xcd0 = [order[b] for b in range(0, len(order), 8) if order[b] is not None]
    if nfull * G >= 2 * G and per >= 2 * G:
        pts = [(blk - ((blk * 64) // P) * nfull - (((blk * 64) // P) * r + 63) // 64) for blk in xcd0[:G]]
        assert len(set(pts)) == 1   # the first G workgroups of XCD 0: one point tile, G groups


def padded_order(L, V, bpg, live):
    """blockIdx -> block (or ('zero', block) for a surplus workgroup that zero-fills an all-padding block, or None) of the padded
    launch: groups of `bpg` blocks, the first L of which hold `live` real rows; the V dumped groups are evaluated in full."""
    groups, all_ = L, L + V
    rb = min((live + 63) // 64, bpg)
    real, tail = groups * rb, V * bpg
    btot = real + tail
    per = (btot + 7) // 8
    first_surplus = 8 * per
    grid = groups * bpg + tail + 7
    out = []
    for b in range(grid):
        if b < first_surplus:
            v = (b & 7) * per + (b >> 3)
            if v >= btot:
                out.append(None)
                continue
            head = rb * all_
            if v < head:
                pt, l = divmod(v, all_)
            else:
                q, rem = divmod(v - head, V)
                pt, l = rb + q, groups + rem
            out.append(l * bpg + pt)
        else:
            s, db = b - first_surplus, bpg - rb
            out.append(('zero', (s // db) * bpg + rb + s % db) if (db > 0 and s < groups * db) else None)
    return out, rb


@pytest.mark.parametrize('L,V,bpg,live', [(5, 2, 16, 517), (96, 8, 64, 3686), (3, 1, 4, 0), (9, 8, 16, 1024), (11, 3, 64, 3700), (5, 2, 10, 64),
                                         (96, 8, 512, 29487), (2, 1, 1, 64), (2, 1, 1, 1)])
def test_padded_point_major_order_covers_every_block_once(L, V, bpg, live):
    order, rb = padded_order(L, V, bpg, live)
    evaluated = [b for b in order if isinstance(b, int)]
    zeroed = [b[1] for b in order if isinstance(b, tuple)]
    want_eval = [l * bpg + pt for l in range(L) for pt in range(rb)] + [l * bpg + pt for l in range(L, L + V) for pt in range(bpg)]
    want_zero = [l * bpg + pt for l in range(L) for pt in range(rb, bpg)]
    assert sorted(evaluated) == sorted(want_eval)
    assert sorted(zeroed) == sorted(want_zero)
    assert len(set(evaluated) & set(zeroed)) == 0
