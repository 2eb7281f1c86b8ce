"""The one-pass delete (minivectordb_amd/csrc/util_kernels.hpp: shift_save_kernel + shift_rows_kernel; host side mvdb.hip:
mvdb_index_remove_rows), restated on the CPU.  The tail of the matrix is shifted in place in 16-byte units: every workgroup owns
a contiguous range of the NEW tail and walks it upwards in slices (sources into registers, barrier, store); sources that lie in
the NEXT workgroup's range come from a side copy taken before the launch.  The emulation runs the workgroups in adversarial
orders — last to first, interleaved slice by slice, random — which is exactly what the side copies must make harmless, and
compares with np.delete.  (On the GPU: tests/test_flat_gpu.py::test_remove_a_few_rows_compacts_in_one_pass.)
"""
import numpy as np
import pytest

SLICE = 256 * 8     # units per workgroup and slice (kShiftUnits = 8)


def _plan(new_units, side_units, cus=4, side_cap=1 << 30):
    """mvdb_index_remove_rows: groups, range (host arithmetic restated)."""
    groups = min(cus * 8, (new_units + SLICE - 1) // SLICE)
    fit = side_cap // (side_units * 16) + 1
    groups = max(1, min(groups, fit))
    rng = ((new_units + groups - 1) // groups + SLICE - 1) // SLICE * SLICE
    groups = (new_units + rng - 1) // rng
    return groups, rng


def _shift_of(dels_rel, r):
    """smallest j with del[j] - j > r"""
    lo, hi = 0, len(dels_rel)
    while lo < hi:
        mid = (lo + hi) // 2
        if dels_rel[mid] - mid > r:
            hi = mid
        else:
            lo = mid + 1
    return lo


def _emulate(tail_units, dels_rel, rowunits, order, cus=4):
    """tail_units: 1-D array, one entry per 16-byte unit of the OLD tail (starting at the first deleted row)."""
    m = len(dels_rel)
    old_units = tail_units.size
    new_units = old_units - m * rowunits
    side_units = m * rowunits
    if new_units == 0:                               # the deleted rows were the last ones: nothing moves (the host skips the launch)
        return tail_units[:0]
    groups, rng = _plan(new_units, side_units, cus)
    tail = tail_units.copy()
    # shift_save_kernel: side[b][j] = tail[(b + 1) range + j]
    side = np.full((max(groups - 1, 1), side_units), -1, dtype=tail.dtype)
    for b in range(groups - 1):
        base = (b + 1) * rng
        cnt = max(0, min(side_units, old_units - base))
        side[b, :cnt] = tail[base:base + cnt]
    # shift_rows_kernel, one (group, slice) step at a time in the given order
    steps = [(g, p) for g in range(groups) for p in range(g * rng, min(g * rng + rng, new_units), SLICE)]
    for g, p in order(steps, groups):
        a, bound = g * rng, g * rng + rng
        b = min(bound, new_units)
        has_next = g + 1 < groups
        u = np.arange(p, min(p + SLICE, b))
        lo = np.array([_shift_of(dels_rel, int(r)) for r in np.unique(u // rowunits)])
        lo_u = lo[np.searchsorted(np.unique(u // rowunits), u // rowunits)]
        s = u + lo_u * rowunits
        from_side = has_next & (s >= bound)
        v = np.where(from_side, side[min(g, side.shape[0] - 1), np.clip(s - bound, 0, side_units - 1)], tail[np.clip(s, 0, old_units - 1)])
        tail[u] = v                                  # (loads of the slice all happened above: the barrier)
    return tail[:new_units]


def _in_order(steps, groups):
    return steps


def _last_group_first(steps, groups):
    return sorted(steps, key=lambda gp: (-gp[0], gp[1]))


def _random(steps, groups):
    rs = np.random.RandomState(1)
    # a random interleaving that keeps every group's own slices in ascending order (a workgroup walks upwards)
    per = {g: [s for s in steps if s[0] == g] for g in range(groups)}
    out = []
    live = [g for g in per if per[g]]
    while live:
        g = live[rs.randint(len(live))]
        out.append(per[g].pop(0))
        if not per[g]:
            live.remove(g)
    return out


@pytest.mark.parametrize("order", [_in_order, _last_group_first, _random])
@pytest.mark.parametrize("rows,rowunits,dels", [
    (5000, 16, [0]), (5000, 16, [7]), (5000, 16, [4999]), (3000, 32, [10, 11, 12, 13]), (4000, 16, [3, 400, 401, 3999]),
    (9000, 8, [0, 1, 2, 5, 8, 13, 21, 8000]), (2000, 128, [1]), (70000, 4, [5]), (70000, 4, [2, 3, 60000]),
])
def test_in_place_shift_equals_np_delete_in_any_workgroup_order(rows, rowunits, dels, order):
    first = min(dels)
    x = np.arange(rows * rowunits, dtype=np.int64).reshape(rows, rowunits)        # every unit its own value
    want = np.delete(x, dels, 0)[first:].reshape(-1)
    dels_rel = sorted(d - first for d in dels)
    got = _emulate(x[first:].reshape(-1), dels_rel, rowunits, order)
    assert got.size == want.size
    bad = np.flatnonzero(got != want)
    assert bad.size == 0, (bad[:5], got[bad[:5]], want[bad[:5]])
