"""The depth sort by counting (csrc/binning.hip) restated in numpy, kernel by kernel -- digit layout from
the varying key bits, per-block bucket counts, column scan, a scatter that hands the slots of a (block, bucket) group out
in ARBITRARY order, ranges of whole buckets cut at multiples of 2048, the relative key, the distribution sort with its
(key, instance number) tie-break and the passes over (instance number, key) for clustered ranges -- and held against a stable
argsort of the keys: the order of the visible instances must be that one, whatever the scatter's arbitrary choices were.
Pins the ALGORITHM on the CPU; the GPU suite pins the kernels (test_depth_sort_by_counting_equals_the_look_back_passes)."""
import zlib

import numpy as np
import pytest

MSD_BITS, RANGE, CAP, DIST_BITS, DIST_MAX, DIGIT = 12, 2048, 4096, 11, 16, 9
CULLED = 0xFFFFFFFF


def layout(keys_visible):
    """DepthLayout of binning.hip: the window [lo, lo + nbits) of the bits that vary among the visible keys."""
    if keys_visible.size == 0:
        return 0, 1
    o = int(np.bitwise_or.reduce(keys_visible))
    nz = int(np.bitwise_or.reduce(~keys_visible & 0xFFFFFFFF))
    varying = o & nz
    if varying == 0:
        return 0, 1
    lo = (varying & -varying).bit_length() - 1
    return lo, varying.bit_length() - lo


def digits(nbits):
    """(shift, width) of the <= 9-bit digits a key of `nbits` bits is sorted by, least significant first."""
    n = -(-nbits // DIGIT)
    base, rem = divmod(nbits, n)
    return [(p * base + min(p, rem), base + (p < rem)) for p in range(n)]


def depth_sort_by_counting(keys, tile, rng, cap=CAP, dist_max=DIST_MAX):
    """keys: u32 per instance (CULLED = no pairs).  Returns (inst_sorted, ranges taken off chip, ranges sorted by passes)."""
    keys = np.asarray(keys, np.uint32).astype(np.int64)
    I = keys.size
    vis = keys != CULLED
    lo, nbits = layout(keys[vis].astype(np.uint32))
    w = min(MSD_BITS, nbits)
    shift, mask = lo + nbits - w, (1 << w) - 1
    bucket = (keys >> shift) & mask
    rows = -(-I // tile)
    # 1. depth_msd_count_kernel: counts[row][bucket], culled per row
    counts = np.zeros((rows, 1 << MSD_BITS), np.int64)
    blk = np.arange(I) // tile
    np.add.at(counts, (blk[vis], bucket[vis]), 1)
    culled_rows = np.bincount(blk[~vis], minlength=rows)
    # 2. depth_msd_colscan_kernel
    bases = np.cumsum(counts, axis=0) - counts
    totals = counts.sum(axis=0)
    starts = np.concatenate([[0], np.cumsum(totals)])           # starts[4096] = n_vis
    n_vis = int(starts[-1])
    # 3. depth_msd_scatter_kernel: slots of a (block, bucket) group in arbitrary order; culled straight to the end
    out_key, out_inst = np.full(n_vis, -1, np.int64), np.full(n_vis, -1, np.int64)
    inst_sorted = np.full(I, -1, np.int64)
    for b in range(rows):
        idx = np.arange(b * tile, min(I, (b + 1) * tile))
        v = idx[vis[idx]]
        v = v[rng.permutation(v.size)]                          # whatever order the LDS atomics serve the lanes in
        taken = {}
        for i in v:
            d = int(bucket[i])
            slot = taken.get(d, 0)
            taken[d] = slot + 1
            p = starts[d] + bases[b, d] + slot
            out_key[p], out_inst[p] = keys[i], i
        c = idx[~vis[idx]]
        first = n_vis + int(culled_rows[:b].sum())
        inst_sorted[first:first + c.size] = c                   # in index order
    assert (out_inst >= 0).all()
    # 4. depth_range_sort_kernel
    low = shift - lo
    wmask = (1 << nbits) - 1
    ibits = max(1, int(I - 1).bit_length())
    off_chip = by_passes = 0
    for k in range(-(-I // RANGE) + 1):
        t0, t1 = min(n_vis, k * RANGE), min(n_vis, k * RANGE + RANGE)
        ge0, ge1 = np.flatnonzero(starts[:-1] >= t0), np.flatnonzero(starts[:-1] >= t1)
        r0 = int(starts[ge0[0]]) if ge0.size else n_vis
        r1 = int(starts[ge1[0]]) if ge1.size else n_vis
        b1 = int(ge1[0]) if ge1.size else 1 << MSD_BITS
        if r0 >= r1:
            continue
        b0 = int(np.flatnonzero(starts[:-1] == r0)[-1])        # the last bucket starting at r0: the first occupied one
        assert totals[b0] > 0 and starts[b0] == r0
        rel = ((out_key[r0:r1] >> lo) & wmask) - (b0 << low)
        span = (b1 - b0) << low
        assert (rel >= 0).all() and (rel < span).all()
        inst = out_inst[r0:r1]
        rbits = max(1, int(span - 1).bit_length()) if span > 1 else 1
        n = r1 - r0
        if n <= cap:
            dsh = max(0, rbits - DIST_BITS)
            hist = np.bincount(rel >> dsh, minlength=1 << DIST_BITS)
            if hist.max() <= dist_max:
                # distribution sort: bucket of the relative key's top bits, then (key, instance number) inside the bucket
                first = np.cumsum(hist) - hist
                res = np.empty(n, np.int64)
                for j in range(n):
                    d = rel[j] >> dsh
                    mates = np.flatnonzero((rel >> dsh) == d)
                    r = int(((rel[mates] < rel[j]) | ((rel[mates] == rel[j]) & (inst[mates] < inst[j]))).sum())
                    res[first[d] + r] = inst[j]
                inst_sorted[r0:r1] = res
                continue
            by_passes += 1
        else:
            off_chip += 1
        # stable passes, least significant digit first: over the instance numbers, then over the relative keys (on chip and
        # off chip the same passes; off chip they run chunk by chunk through memory)
        order = np.arange(n)
        for sh, wd in digits(ibits):
            order = order[np.argsort((inst[order] >> sh) & ((1 << wd) - 1), kind="stable")]
        for sh, wd in digits(rbits):
            order = order[np.argsort((rel[order] >> sh) & ((1 << wd) - 1), kind="stable")]
        inst_sorted[r0:r1] = inst[order]
    return inst_sorted, off_chip, by_passes


def reference(keys):
    """What the look-back passes leave for the visible instances: a stable sort by key; then the culled ones by index."""
    keys = np.asarray(keys, np.uint32).astype(np.int64)
    vis = np.flatnonzero(keys != CULLED)
    return np.concatenate([vis[np.argsort(keys[vis], kind="stable")], np.flatnonzero(keys == CULLED)])


def depth_keys(rng, n, zlo=2.0, zhi=10.0):
    return rng.uniform(zlo, zhi, n).astype(np.float32).view(np.uint32)


@pytest.mark.parametrize("case", ["uniform", "culled", "duplicates", "one_key", "clustered", "narrow", "small_cap", "tiny"])
def test_counting_depth_sort_is_the_stable_sort(case):
    rng = np.random.default_rng(zlib.crc32(case.encode()))
    tile, kw = 1024, {}
    if case == "uniform":
        keys = depth_keys(rng, 30_000)
    elif case == "culled":                      # nine tenths culled: they must not take part
        keys = depth_keys(rng, 20_000)
        keys[rng.random(20_000) < 0.9] = CULLED
    elif case == "duplicates":                  # every key four times: equal keys in input order
        keys = np.repeat(depth_keys(rng, 5_000), 4)[rng.permutation(20_000)]
    elif case == "one_key":                     # no varying bit: one bucket, one range that cannot fit
        keys = np.full(9_000, np.float32(5.0).view(np.uint32), np.uint32)
    elif case == "clustered":                   # a thousand instances within a few ulp: a range that takes the passes
        keys = depth_keys(rng, 12_000)
        keys[:1000] = np.float32(5.0).view(np.uint32) + rng.integers(0, 4, 1000).astype(np.uint32)
        keys = keys[rng.permutation(12_000)]
    elif case == "narrow":                      # nine varying bits only: the first digit IS the key
        keys = (np.float32(4.0).view(np.uint32) + rng.integers(0, 512, 6_000)).astype(np.uint32)
    elif case == "small_cap":                   # ranges above 64 elements go off chip
        keys, kw = depth_keys(rng, 8_000), {"cap": 64}
    else:
        keys = depth_keys(rng, 37)
    got, off_chip, by_passes = depth_sort_by_counting(keys, tile, rng, **kw)
    assert np.array_equal(got, reference(keys))
    if case == "one_key":
        assert off_chip == 1
    if case == "clustered":
        assert by_passes >= 1 and off_chip == 0
    if case == "small_cap":
        assert off_chip >= 3        # four ranges of ~2000 instances
    if case == "uniform":
        assert off_chip == 0 and by_passes == 0
