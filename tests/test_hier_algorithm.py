"""The hierarchical tile sort (csrc/binning.hip) restated in numpy, kernel by kernel -- elements per (8 x 8-tile
super-tile, instance) with the clipped rectangle packed above the key, ONE stable sort by key, XCD-class-major chunk numbering,
per-(chunk, wave, tile) counts from the row / column cover masks, tile scan in tile-id order, scatter by set bits in element
order -- and held against the oracle's duplicateWithKeys + 64-bit stable sort on the same frame: `point_list` and `ranges`
bit for bit.  Pins the ALGORITHM on the CPU (the GPU suite pins the kernels); it is also what told a compiler-invisible
hazard (v_cmp -> v_writelane wait states) apart from a logic error when the first GPU build sorted wrongly."""
import numpy as np
import pytest

import helpers as Hh
from casualhdrsplat_amd import synthetic as S

SUPER = 8


def hier_sort(x0, y0, w, h, order, gx, gy, chunk):
    """Rectangles (tiles) of the instances, `order` = instances by depth.  Returns (point_list, ranges[tile] = (first, end))."""
    sgx, sgy = (gx + SUPER - 1) // SUPER, (gy + SUPER - 1) // SUPER
    nst = sgx * sgy
    kb = int(nst).bit_length()
    words, insts = [], []
    for i in order:                                             # hier_emit_kernel: elements in depth order, row-major
        if w[i] == 0 or h[i] == 0:
            continue
        sx0, sy0 = x0[i] // SUPER, y0[i] // SUPER
        sw, sh = (x0[i] + w[i] - 1) // SUPER - sx0 + 1, (y0[i] + h[i] - 1) // SUPER - sy0 + 1
        for t in range(sw * sh):
            cy, cx = divmod(t, sw)
            key = (sy0 + cy) * sgx + sx0 + cx
            ox, oy = (sx0 + cx) * SUPER, (sy0 + cy) * SUPER
            lx0, lx1 = max(x0[i], ox) - ox, min(x0[i] + w[i], ox + SUPER) - ox
            ly0, ly1 = max(y0[i], oy) - oy, min(y0[i] + h[i], oy + SUPER) - oy
            words.append(key | ((lx0 | (ly0 << 3) | ((lx1 - lx0 - 1) << 6) | ((ly1 - ly0 - 1) << 9)) << kb))
            insts.append(i)
    words, insts = np.array(words, np.int64), np.array(insts, np.int64)
    keys = words & ((1 << kb) - 1)
    srt = np.argsort(keys, kind="stable")                       # the radix pass(es): stable, by key only
    words, insts, keys = words[srt], insts[srt], keys[srt]
    cnt = np.bincount(keys, minlength=nst)
    first = np.concatenate([[0], np.cumsum(cnt)])
    desc, per = [], (nst + 7) // 8                              # hier_plan_kernel: chunks numbered XCD-class-major
    for x in range(8):
        for j in range(per):
            s = j * 8 + x
            if s < nst:
                for k in range((cnt[s] + chunk - 1) // chunk):
                    desc.append((s, first[s] + k * chunk, min(first[s] + cnt[s], first[s] + (k + 1) * chunk), k))

    def cover(c_words):                                         # hier_cover: bit j of mask[tile] = element j covers the tile
        r = c_words >> kb
        lx0, ly0, lw, lh = r & 7, (r >> 3) & 7, ((r >> 6) & 7) + 1, ((r >> 9) & 7) + 1
        ty, tx = np.arange(64)[:, None] >> 3, np.arange(64)[:, None] & 7
        return (ly0 <= ty) & (ty < ly0 + lh) & (lx0 <= tx) & (tx < lx0 + lw)          # [64 tiles, elements]

    wave = chunk // 4
    counts = np.zeros((len(desc), 4, 64), np.int64)             # hier_count_kernel
    total = np.zeros(nst * 64, np.int64)
    for c, (s, b, e, k) in enumerate(desc):
        for wv in range(4):
            wb = min(e, b + wv * wave)
            we = min(e, wb + wave)
            if we > wb:
                counts[c, wv] = cover(words[wb:we]).sum(axis=1)
        total[s * 64:(s + 1) * 64] += counts[c].sum(axis=0)
    start = np.zeros(nst * 64, np.int64)                        # hier_tiles_kernel: scan in tile-id order
    ranges = np.zeros((gx * gy, 2), np.int64)
    run = 0
    for t in range(gx * gy):
        ty, tx = divmod(t, gx)
        at = ((ty // SUPER) * sgx + tx // SUPER) * 64 + (ty % SUPER) * 8 + tx % SUPER
        start[at] = run
        if total[at]:
            ranges[t] = (run, run + total[at])
        run += total[at]
    pl = np.full(run, -1, np.int64)                             # hier_scatter_kernel
    for c, (s, b, e, k) in enumerate(desc):
        earlier = counts[c - k:c].sum(axis=(0, 1)) if k else np.zeros(64, np.int64)
        for wv in range(4):
            pos = start[s * 64:(s + 1) * 64] + earlier + counts[c, :wv].sum(axis=0)
            wb = min(e, b + wv * wave)
            we = min(e, wb + wave)
            for r0 in range(wb, we, 64):                        # 64 elements at a time, lane = tile walks its set bits in order
                m = cover(words[r0:min(we, r0 + 64)])
                for t in range(64):
                    js = np.nonzero(m[t])[0]
                    pl[pos[t]:pos[t] + len(js)] = insts[r0 + js]
                    pos[t] += len(js)
    return pl, ranges


@pytest.mark.parametrize("P,W,H,seed,chunk,skew", [(3000, 500, 300, 3, 512, False), (3000, 500, 300, 4, 1024, False),
                                                   (4000, 640, 384, 5, 512, True), (800, 37, 23, 6, 512, False)])
def test_hierarchical_tile_sort_restated_in_numpy_matches_the_oracle(oracle, P, W, H, seed, chunk, skew):
    sc = S.make_scene(P, W, H, 0, seed=seed)
    if skew:       # half the cloud inside one super-tile, some Gaussians huge: many chunks in one key, rectangles of 64 tiles
        half = np.arange(P) % 2 == 0
        sc.means3D[half, 0] *= 0.15
        sc.means3D[half, 1] *= 0.15
        sc.scales[::97] *= 25.0
    f, _ = Hh.run_oracle(oracle, sc, backward=False)
    gx, gy = (W + 15) // 16, (H + 15) // 16
    # the oracle's rectangles: recovered from its pair list (tiles of every instance), depth order from its keys
    R = f["R"]
    tiles = (f["keys_sorted"] >> np.uint64(32)).astype(np.int64)
    inst = f["point_list"].astype(np.int64)
    x0 = np.zeros(P, np.int64); y0 = np.zeros(P, np.int64); w = np.zeros(P, np.int64); h = np.zeros(P, np.int64)
    for i in np.unique(inst):
        t = tiles[inst == i]
        ty, tx = t // gx, t % gx
        x0[i], y0[i], w[i], h[i] = tx.min(), ty.min(), tx.max() - tx.min() + 1, ty.max() - ty.min() + 1
    assert int((w * h).sum()) == R and np.array_equal((w * h)[: P], f["tiles_touched"].astype(np.int64))
    order = np.lexsort((np.arange(P), Hh.bits(f["depths"]).astype(np.int64)))      # by depth bits, ties by instance
    pl, ranges = hier_sort(x0, y0, w, h, order, gx, gy, chunk)
    assert np.array_equal(pl, inst)
    assert np.array_equal(ranges, f["ranges"].astype(np.int64))
