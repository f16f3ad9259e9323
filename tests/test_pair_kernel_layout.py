"""Host-side restatement of raster_fwd_pair_kernel's lane -> pixel layout and of the lanes its block bounds are read from
(gsx_raster_fast.hip: two pixels per lane, eight 4x4 blocks per wave, two waves per 16x16 tile).  The kernel takes the (u, v)
rectangle of a block from the lanes that own the block's first and last VALID pixel — wave-uniform lane numbers computed from the
image size — instead of reducing over the lanes; this test checks those lane numbers against a brute-force search for every ragged
width / height, and that the layout covers a tile exactly once.  (The GPU side is tests/test_gpu_fused.py::test_forward_kernels_are_bit_identical.)"""
import itertools

TILE = 16


def lane_pixels(wave, lane):
    """(row, col0), (row, col0 + 1) of the tile for this lane — the kernel's j0 / i formulas."""
    q8, k8 = lane >> 3, lane & 7
    col0 = (q8 & 3) * 4 + (k8 & 1) * 2
    row = wave * 8 + (q8 >> 2) * 4 + (k8 >> 1)
    return (row, col0), (row, col0 + 1)


def test_layout_covers_the_tile_once_and_blocks_are_4x4():
    seen = {}
    for wave, lane in itertools.product(range(2), range(64)):
        for p, (r, c) in enumerate(lane_pixels(wave, lane)):
            assert (r, c) not in seen
            seen[(r, c)] = (wave, lane, p)
    assert len(seen) == TILE * TILE
    for wave, q8 in itertools.product(range(2), range(8)):   # the eight lanes of a block own one 4x4 block: column q8 & 3, row q8 >> 2 of the wave's half
        px = [rc for lane in range(8 * q8, 8 * q8 + 8) for rc in lane_pixels(wave, lane)]
        rows, cols = {r for r, _ in px}, {c for _, c in px}
        assert rows == set(range(wave * 8 + (q8 >> 2) * 4, wave * 8 + (q8 >> 2) * 4 + 4))
        assert cols == set(range((q8 & 3) * 4, (q8 & 3) * 4 + 4))


def test_bound_lanes_own_the_first_and_last_valid_pixel():
    for W, H, wave in itertools.product(range(1, TILE + 1), range(1, TILE + 1), range(2)):   # valid columns / rows of a (last) tile
        for k in range(4):   # block column k: u-range from lanes of block row 0 (q8 = k)
            last = min(3, W - 1 - 4 * k)
            valid = [c for c in range(4 * k, 4 * k + 4) if c < W]
            assert (last >= 0) == bool(valid)
            if valid:
                ll = max(last, 0)
                first_lane, last_lane, last_p = 8 * k, 8 * k + (ll >> 1), ll & 1
                assert lane_pixels(wave, first_lane)[0][1] == valid[0]
                assert lane_pixels(wave, last_lane)[last_p][1] == valid[-1]
        for k in range(2):   # block row k: v-range from lanes of block column 0 (q8 = 4 k)
            y0 = wave * 8 + 4 * k
            last = min(3, H - 1 - y0)
            valid = [r for r in range(y0, y0 + 4) if r < H]
            assert (last >= 0) == bool(valid)
            if valid:
                ll = max(last, 0)
                assert lane_pixels(wave, 32 * k)[0][0] == valid[0]
                assert lane_pixels(wave, 32 * k + 2 * ll)[0][0] == valid[-1]
