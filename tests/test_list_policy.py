"""Host-side policies of the fused render path that need no GPU: the list granularity of the next frame (16- or 32-pixel lists, with a
hysteresis band) and the layout of the backward's records (chains or per-Gaussian ranges), both chosen from the previous frame's counts."""
import importlib

import pytest


@pytest.fixture()
def rz(monkeypatch):
    import gsx  # noqa: F401
    rasterizer = importlib.import_module("gsx.rasterizer")
    monkeypatch.delenv("GSX_LIST_TILE", raising=False)
    monkeypatch.setattr(rasterizer, "_LIST_TILE_STATE", {})
    monkeypatch.setattr(rasterizer, "_ENTRIES_PER_GAUSSIAN", {})
    return rasterizer


def test_list_tile_hysteresis_and_record_ranges(rz):
    key = (1296, 840, 0)
    tiles16, tiles32 = 81 * 53, 41 * 27
    assert rz._list_tile_for(key) == 16 and not rz._record_ranges_for(key)
    # a light frame: 400 entries per tile, 3 per Gaussian -> 16-pixel lists, chained records
    rz._list_tile_update(key, 16, 400 * tiles16, tiles16, 400 * tiles16 // 3)
    assert rz._list_tile_for(key) == 16 and not rz._record_ranges_for(key)
    # dense in entries per Gaussian but not per tile (the trained phase of the garden stand-in): still 16-pixel lists, records in ranges
    rz._list_tile_update(key, 16, 2000 * tiles16, tiles16, 2000 * tiles16 // 12)
    assert rz._list_tile_for(key) == 16 and rz._record_ranges_for(key)
    # heavy per tile: the next frame gets 32-pixel lists (and ranges with them)
    rz._list_tile_update(key, 16, int(rz.LIST_TILE_UP) * tiles16, tiles16, 1_000_000)
    assert rz._list_tile_for(key) == 32 and rz._record_ranges_for(key)
    # inside the band nothing flips ...
    rz._list_tile_update(key, 32, int((rz.LIST_TILE_DOWN + 200) * tiles32), tiles32, 1_000_000)
    assert rz._list_tile_for(key) == 32
    # ... below it the lists go back to 16 pixels; the records follow the entries per Gaussian
    rz._list_tile_update(key, 32, int((rz.LIST_TILE_DOWN - 200) * tiles32), tiles32, 1_000_000)
    assert rz._list_tile_for(key) == 16 and not rz._record_ranges_for(key)
    # another image shape has its own state
    assert rz._list_tile_for((1920, 1080, 0)) == 16 and not rz._record_ranges_for((1920, 1080, 0))


def test_list_tile_switch_for_tests(rz, monkeypatch):
    monkeypatch.setenv("GSX_LIST_TILE", "32")
    assert rz._list_tile_for((64, 64, 0)) == 32 and rz._record_ranges_for((64, 64, 0))
    monkeypatch.setenv("GSX_LIST_TILE", "16")
    assert rz._list_tile_for((64, 64, 0)) == 16
