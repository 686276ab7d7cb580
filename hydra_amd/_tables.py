"""
Host-side index tables for the HIP kernels (integer bookkeeping only).

``baseline_indices`` turns the reference's (ants, antpairs) description into index
arrays; ``build_segments`` groups baselines by antenna block so that the voltages
of one block pair fit in the 160 KiB LDS of a gfx950 CU for the fused Gram kernel
(csrc/gram.hip: ``gram_fused_kernel``).
"""
import numpy as np

LDS_BYTES = 160 * 1024
ROW_BYTES = 256 * 16          # one antenna: 128 + 128 sources, complex128
INFO_BYTES = 12               # per baseline row: packed LDS rows (4) + sqrt weight (8)


_BL_CACHE = {}


def _fingerprint(antpairs, ants):
    """Cheap content check of a cached (antpairs, ants) pair: length plus the first, the last
    and ~30 strided pairs, and all antenna names.  A caller that shuffles, reorders or replaces
    pairs in place changes it (identity alone would return stale indices silently)."""
    n = len(antpairs)
    step = max(1, n // 29)
    picks = tuple((antpairs[k][0], antpairs[k][1]) for k in range(0, n, step))
    last = (antpairs[-1][0], antpairs[-1][1]) if n else ()
    return n, picks, last, tuple(np.atleast_1d(np.asarray(ants)).tolist())


def baseline_indices(ants, antpairs):
    """Index of each baseline's antennas in ``ants`` (the lookup of
    ptsrc_sampler.py:53-56 / utils.py:102-105).  Returns int32 arrays (bl_i, bl_j).
    The result is cached per (antpairs, ants) object pair and re-validated by a content
    fingerprint on every hit: the heads call this every Gibbs iteration with the same lists
    (54 615 pairs for HERA-331)."""
    key = (id(antpairs), id(ants))
    hit = _BL_CACHE.get(key)
    if hit is not None and hit[0] is antpairs and hit[1] is ants \
            and hit[2] == _fingerprint(antpairs, ants):
        return hit[3], hit[4]
    ants_l = [a for a in np.atleast_1d(np.asarray(ants)).tolist()]
    pos = {}
    for k, a in enumerate(ants_l):
        pos.setdefault(a, k)
    try:
        bl_i = np.array([pos[p[0]] for p in antpairs], dtype=np.int32)
        bl_j = np.array([pos[p[1]] for p in antpairs], dtype=np.int32)
    except KeyError as exc:
        raise IndexError(f"antenna {exc.args[0]} of `antpairs` is not in `ants`") from exc
    if len(_BL_CACHE) > 16:
        _BL_CACHE.clear()
    _BL_CACHE[key] = (antpairs, ants, _fingerprint(antpairs, ants), bl_i, bl_j)
    return bl_i, bl_j


def is_complete_array(bl_i, bl_j, nant):
    """True if the baselines are exactly the nant(nant-1)/2 pairs i<j, each once (any order)."""
    bl_i = np.asarray(bl_i, dtype=np.int64)
    bl_j = np.asarray(bl_j, dtype=np.int64)
    if bl_i.size != nant * (nant - 1) // 2 or nant < 2:
        return False
    if not np.all(bl_i < bl_j):
        return False
    return np.unique(bl_i * nant + bl_j).size == bl_i.size


def build_segments(bl_i, bl_j, nant):
    """Segment and row tables for ``hydra_gram_fused``.

    Returns (seg, rowtab): seg (nseg, 6) int32 rows [i0, ni, j0, nj, row0, nrows];
    rowtab (nbl, 3) int32 rows [LDS row of antenna i, LDS row of antenna j, baseline b].
    With few antennas one segment holds them all; otherwise antennas are cut into blocks of
    ``ab`` and baselines are grouped by (block of i, block of j).
    """
    bl_i = np.asarray(bl_i, dtype=np.int64)
    bl_j = np.asarray(bl_j, dtype=np.int64)
    nbl = bl_i.size
    assert bl_j.size == nbl and nbl > 0
    assert bl_i.min() >= 0 and bl_j.min() >= 0 and max(bl_i.max(), bl_j.max()) < nant
    if nant * ROW_BYTES + nbl * INFO_BYTES + 16 <= LDS_BYTES and nant < 65536:
        seg = np.array([[0, nant, 0, nant, 0, nbl]], dtype=np.int32)
        rowtab = np.stack([bl_i, bl_j, np.arange(nbl)], axis=1).astype(np.int32)
        return seg, np.ascontiguousarray(rowtab)
    ab = 19                       # 2*19 antennas * 4 KiB + 19^2 rows * 12 B < 160 KiB
    bi, bj = bl_i // ab, bl_j // ab
    order = np.lexsort((bl_j, bl_i, bj, bi))
    segs, rows = [], []
    row0 = 0
    keys = np.stack([bi[order], bj[order]], axis=1)
    start = 0
    for k in range(1, nbl + 1):
        if k == nbl or (keys[k] != keys[start]).any():
            I, J = int(keys[start, 0]), int(keys[start, 1])
            i0, j0 = I * ab, J * ab
            ni, nj = min(ab, nant - i0), min(ab, nant - j0)
            sel = order[start:k]
            li = bl_i[sel] - i0
            lj = bl_j[sel] - j0 + (0 if I == J else ni)
            assert (ni + (0 if I == J else nj)) * ROW_BYTES + sel.size * INFO_BYTES + 16 <= LDS_BYTES
            segs.append([i0, ni, j0, nj, row0, sel.size])
            rows.append(np.stack([li, lj, sel], axis=1))
            row0 += sel.size
            start = k
    seg = np.asarray(segs, dtype=np.int32)
    rowtab = np.ascontiguousarray(np.concatenate(rows, axis=0).astype(np.int32))
    return seg, rowtab
