"""CPU models of the multi-block sorts of the binned intersection (gsx_intersect.hip), statement for statement, so that their index
arithmetic is checked where no GPU is needed: (1) giant segments — 16 384-key sorted chunks, then merge passes that are parallel over
16 384-key output windows whose merge-path splits a wave finds by a 64-way search (giant_list / giant_chunk_sort / giant_merge,
merge_split_wave); (2) the ranked variant's bitmap sort (tile_sort_bitmap_kernel: per-wave word ranges, popcount prefix, emission); (3) its depth ranks: the
four-pass LSD radix sort (block histograms, per-digit scan over the blocks, ballot-ranked stable scatter).
The GPU kernels themselves are compared bit for bit with the device-wide sort in tests/test_gpu_ops.py."""
import numpy as np
import pytest

CAP = 16384  # TSORT_BIG_CAP


def merge_split_wave(A, B, d):
    """number of outputs among the first d that come from A (64 lanes probe 64 points per step; gsx_intersect.hip: merge_split_wave)"""
    la, lb = len(A), len(B)
    lo, hi = max(0, d - lb), min(d, la)
    steps = 0
    while lo < hi:
        span = hi - lo
        mids = [lo + span * (lane + 1) // 65 for lane in range(64)]
        below = [A[m] <= B[d - 1 - m] for m in mids]
        cnt = sum(below)
        assert below == [True] * cnt + [False] * (64 - cnt)          # monotone predicate: the ballot is a prefix mask
        last_true, first_false = lo + span * cnt // 65, lo + span * (cnt + 1) // 65
        lo, hi = (last_true + 1 if cnt > 0 else lo), (first_false if cnt < 64 else hi)
        steps += 1
    return lo, steps


def giant_sort(keys):
    n = len(keys)
    k = (n + CAP - 1) // CAP
    buf = [np.array(keys), np.empty_like(keys)]
    for c in range(k):                                               # giant_chunk_sort_kernel
        buf[0][c * CAP:(c + 1) * CAP].sort()
    passes = 0
    while (1 << passes) < k:
        passes += 1
    out = np.full(n, -1, dtype=keys.dtype)
    for p in range(passes):                                          # giant_merge_kernel, one launch per pass
        src, dst = buf[p & 1], buf[1 - (p & 1)]
        run = CAP << p
        last = p == passes - 1
        for c in range(k):                                           # window c of the segment = chunk c of the list
            o = c * CAP
            pair0 = o & ~(2 * run - 1)
            la = min(run, n - pair0)
            lb = min(run, n - pair0 - la)
            d0, d1 = o - pair0, min(o - pair0 + CAP, la + lb)
            A, B = src[pair0:pair0 + la], src[pair0 + la:pair0 + la + lb]
            a0, s0 = merge_split_wave(A, B, d0)
            a1, s1 = merge_split_wave(A, B, d1)
            assert max(s0, s1) <= 4                                  # 65-fold narrowing per step
            b0, b1 = d0 - a0, d1 - a1
            merged = np.sort(np.concatenate([A[a0:a1], B[b0:b1]]))   # the LDS merge of the two staged pieces
            assert len(merged) == d1 - d0
            (out if last else dst)[pair0 + d0:pair0 + d1] = merged
    return out if passes else buf[0]


@pytest.mark.parametrize("n", [CAP + 1, 2 * CAP, 2 * CAP + 7, 3 * CAP - 1, 5 * CAP + 123, 8 * CAP, 11 * CAP + 4097])
def test_giant_segment_chunks_and_window_merges(n):
    rng = np.random.default_rng(n)
    keys = rng.permutation(n * 3)[:n].astype(np.int64)               # unique keys, as (depth bits, flatten index) keys are
    assert np.array_equal(giant_sort(keys), np.sort(keys))


def test_merge_split_wave_edges():
    A = np.arange(0, 100, 2)
    B = np.arange(1, 61, 2)
    for d in range(len(A) + len(B) + 1):
        a, _ = merge_split_wave(A, B, d)
        merged = np.sort(np.concatenate([A, B]))[:d]
        assert a == int(np.isin(merged, A).sum())
    assert merge_split_wave(A, np.array([], dtype=A.dtype), 17)[0] == 17     # a run without a partner: pure copy windows
    assert merge_split_wave(np.array([], dtype=A.dtype), B, 9)[0] == 0


def bitmap_sort(ranks, total, stage=480):
    """tile_sort_bitmap_kernel: 16 waves, contiguous word ranges, 256 words per emission step, staged or direct stores"""
    n_words = ((total + 31) // 32 + 4095) // 4096 * 4096
    bits = np.zeros(n_words, dtype=np.uint32)
    np.bitwise_or.at(bits, ranks >> 5, (np.uint32(1) << (ranks & 31).astype(np.uint32)))
    per_wave = n_words // 16
    pop = np.array([bin(int(w)).count("1") for w in bits])
    wave_total = pop.reshape(16, per_wave).sum(1)
    out = np.full(len(ranks), -1, dtype=np.int64)
    staged = direct = 0
    for wave in range(16):
        pos = int(wave_total[:wave].sum())
        for w0 in range(wave * per_wave, (wave + 1) * per_wave, 256):
            cnt = pop[w0:w0 + 256].reshape(4, 64)                    # group u, lane l owns word w0 + 64 u + l
            tot = int(cnt.sum())
            base = np.concatenate([[0], np.cumsum(cnt.sum(1))[:-1]])
            for u in range(4):
                incl = np.cumsum(cnt[u])
                for lane in range(64):
                    j = pos + base[u] + incl[lane] - cnt[u][lane]
                    word, rank0 = int(bits[w0 + 64 * u + lane]), (w0 + 64 * u + lane) << 5
                    while word:
                        b = (word & -word).bit_length() - 1
                        out[j] = rank0 + b
                        j += 1
                        word &= word - 1
            staged, direct = staged + (tot <= stage), direct + (tot > stage)
            pos += tot
    return out, staged, direct


@pytest.mark.parametrize("total,n", [(200_000, 5000), (1_048_576, 20000), (70_000, 60000)])
def test_bitmap_sort_emits_ranks_in_order(total, n):
    rng = np.random.default_rng(total + n)
    ranks = rng.permutation(total)[:n].astype(np.int64)
    out, staged, direct = bitmap_sort(ranks, total)
    assert np.array_equal(out, np.sort(ranks))
    if total == 70_000:
        assert direct > 0                                            # a tile holding most of the frame's Gaussians: the direct-store branch
    else:
        assert staged > 0 and direct == 0


# ---- depth ranks: the four-pass LSD radix sort (rs_hist / rs_scan / rs_scatter, gsx_intersect.hip) --------------------------------
RS_BLOCK, RS_ROUNDS, RS_BITS = 256, 8, 8
RS_TILE, RS_BINS = RS_BLOCK * RS_ROUNDS, 1 << RS_BITS


def radix_pass(keys, vals, shift):
    """One digit pass, statement for statement: block histograms hist[digit][block]; per digit an exclusive scan over the blocks + the
    digit total; per block the scan of the totals, then per wave (64-pair rounds in index order) the rank of a pair among the lanes
    below it with the same digit ("match any" by ballots) + the wave's running count + the counts of the waves before it."""
    total = len(keys)
    nblk = (total + RS_TILE - 1) // RS_TILE
    digit = (keys >> np.uint32(shift)) & np.uint32(RS_BINS - 1)
    hist = np.zeros((RS_BINS, nblk), np.int64)
    for b in range(nblk):
        np.add.at(hist[:, b], digit[b * RS_TILE:(b + 1) * RS_TILE], 1)
    prefix = np.cumsum(hist, axis=1) - hist                      # rs_scan_kernel: exclusive over the blocks, in place
    digit_total = hist.sum(axis=1)
    base = np.cumsum(digit_total) - digit_total                  # every scatter block scans the totals itself
    out_k, out_v = np.empty_like(keys), np.empty_like(vals)
    waves = RS_BLOCK // 64
    for b in range(nblk):
        cnt = np.zeros((waves, RS_BINS), np.int64)
        lrank = {}
        for w in range(waves):
            first = b * RS_TILE + w * 64 * RS_ROUNDS
            for r in range(RS_ROUNDS):
                lanes = [i for i in range(first + r * 64, first + r * 64 + 64) if i < total]
                seen = {}
                for i in lanes:                                   # rank among the valid lanes below with the same digit
                    d = int(digit[i])
                    lrank[i] = cnt[w, d] + seen.get(d, 0)
                    seen[d] = seen.get(d, 0) + 1
                for d, c in seen.items():                         # the highest peer books the round
                    cnt[w, d] += c
        before = np.cumsum(cnt, axis=0) - cnt                     # the counts of the waves before, per digit
        for w in range(waves):
            first = b * RS_TILE + w * 64 * RS_ROUNDS
            for i in range(first, min(first + 64 * RS_ROUNDS, total)):
                d = int(digit[i])
                dst = base[d] + prefix[d, b] + before[w, d] + lrank[i]
                out_k[dst], out_v[dst] = keys[i], vals[i]
    return out_k, out_v


@pytest.mark.parametrize("total,kind", [(1, "random"), (63, "ties"), (2049, "random"), (5000, "ties"), (9000, "high-bits")])
def test_lsd_radix_passes_give_the_stable_order(total, kind):
    rng = np.random.default_rng(total)
    if kind == "random":
        keys = rng.integers(0, 1 << 32, total, dtype=np.uint64).astype(np.uint32)
    elif kind == "ties":
        keys = (rng.integers(1, 40, total).astype(np.float32) * 0.25).view(np.uint32)
    else:
        keys = (rng.integers(0, 4, total, dtype=np.uint64).astype(np.uint32) << np.uint32(30)) | np.uint32(0x00FFFFFF)
    keys[rng.random(total) < 0.3] = 0xFFFFFFFF                    # culled Gaussians rank last
    k, v = keys.copy(), np.arange(total, dtype=np.uint32)
    for p in range(4):
        k, v = radix_pass(k, v, RS_BITS * p)
    want = np.argsort(keys, kind="stable")
    assert np.array_equal(v, want.astype(np.uint32)) and np.array_equal(k, keys[want])
