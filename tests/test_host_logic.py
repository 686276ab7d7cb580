"""
Host-side logic of the multi-GPU solve and of the CG flag schedule, without a GPU: the
block-cyclic row layout (hydra_amd._dist.RowLayout) that `precompute_mpi(shard_rows=True)`,
`RowShardedOperator` and `factor_dense_rows` share, and the helpers that decide when
`linear_solver.cg(linear_op=)` reads the converged flag (they also decide how often a collective
operator is called, so ranks must agree on them: linear_solver.py:466-496 of the reference has
every rank make the same calls).
"""
import numpy as np
import pytest
import torch

from hydra_amd import _dist, linear_solver


@pytest.mark.parametrize("n,size", [(2, 1), (300, 2), (2306, 3), (5000, 2), (65536, 8), (100000, 8),
                                    (100000, 3), (12288, 5)])
def test_row_layout_is_a_partition_in_whole_panels(n, size):
    lays = [_dist.RowLayout(n, size, r) for r in range(size)]
    lay = lays[0]
    # 256 | block: a 256-wide diagonal block of the factorisation never straddles two owners
    assert lay.block % 256 == 0 and lay.block <= _dist.ROW_BLOCK
    allrows = np.concatenate([l.rows() for l in lays])
    assert np.array_equal(np.sort(allrows), np.arange(n))
    for r, l in enumerate(lays):
        rows = l.rows()
        assert l.nloc() == rows.size == l.nloc(r)
        assert np.all(np.diff(rows) > 0)                       # local order = ascending global order
        for b in l.blocks_of():
            g0, g1 = l.block_range(b)
            assert l.owner(b) == r
            l0 = l.local_start(b)
            assert np.array_equal(rows[l0:l0 + g1 - g0], np.arange(g0, g1))
            assert l.local_index(g0) == l0 and l.local_index(g1) == l0 + g1 - g0
        # rows below any global index: what a right-looking step works on
        for g in (0, 1, 255, 256, n // 2, n - 1, n):
            assert l.local_index(g) == int(np.sum(rows < g))
    assert lay.maxloc == max(l.nloc() for l in lays)
    # balance: with cyclic blocks no rank has more than one block over the mean share of the rows
    # below any block boundary (contiguous blocks left the last rank 2 - 1/size times the mean)
    for b in range(lay.nblocks):
        g0 = lay.block_range(b)[0]
        below = [l.nloc() - l.local_index(g0) for l in lays]
        assert max(below) - min(below) <= lay.block


def test_flag_schedule_blocks():
    take = lambda gen, k: [next(gen) for _ in range(k)]
    # no history: fixed blocks; a numpy operator synchronises anyway: one by one
    assert take(linear_solver._block_sizes(0, True), 3) == [linear_solver.CHECK_EVERY] * 3
    assert take(linear_solver._block_sizes(24, False), 3) == [1, 1, 1]
    # history: the last count first, then short blocks
    assert take(linear_solver._block_sizes(24, True), 3) == [24, 2, 2]
    assert take(linear_solver._block_sizes(400, True), 3) == [linear_solver.FIRST_BLOCK_MAX, 25, 25]
    assert linear_solver.FIRST_BLOCK_MAX == 8 * linear_solver.CHECK_EVERY


def test_history_is_kept_per_operator_and_only_for_converged_solves():
    def site_a():
        return lambda v: v

    def site_b():
        return lambda v: 2 * v

    # a closure rebuilt at one call site keeps its key; another call site never shares it
    assert linear_solver._history_key(site_a()) == linear_solver._history_key(site_a())
    assert linear_solver._history_key(site_a()) != linear_solver._history_key(site_b())

    class Op:
        def __call__(self, v):
            return v

    class Op2(Op):
        hydra_history_key = "named"

    assert linear_solver._history_key(Op()) == linear_solver._history_key(Op())
    assert linear_solver._history_key(Op2()) == "named"
    assert linear_solver._history_key(Op(), explicit="mine") == "mine"

    key = ("test-key", 10, 1e-8, False, "op")
    linear_solver._ITER_HISTORY.pop(key, None)
    st = torch.zeros(8, dtype=torch.float64)
    linear_solver._record_iters(key, st, 1000)                 # ran into maxiters: not a prediction
    assert key not in linear_solver._ITER_HISTORY
    st[1] = 1.0
    linear_solver._record_iters(key, st, 24)
    assert linear_solver._ITER_HISTORY.pop(key) == 24
