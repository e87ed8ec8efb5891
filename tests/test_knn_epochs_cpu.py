"""Host logic of the fused kNN route's epoch plan (knn_svc_amd/ops.py: knn_epochs) — no GPU."""
import pytest

from knn_svc_amd import ops


@pytest.mark.parametrize("nq,npc,blocks", [(1500, 30000, 256), (300, 30000, 256), (3000, 30000, 192), (24000, 180000, 256),
                                           (257, 8197, 256), (32768, 262143, 256), (1, 300, 8)])
def test_epochs_cover_every_column_tile_once_in_growing_steps(nq, npc, blocks):
    ep = ops.knn_epochs(nq, npc, blocks)
    gy = -(-npc // 256)
    assert ep[0][0] == 0 and ep[-1][1] == gy
    for (a0, a1), (b0, b1) in zip(ep, ep[1:]):
        assert a1 == b0 and a0 < a1 and b0 < b1
    e0 = ep[0][1]
    assert e0 == gy or 4 <= e0 <= ops.KNN_COLD_TILES_MAX
    assert e0 * 48 <= ops.KNN_FUSED_CAP                       # ~45 survivors per (row, cold tile) fit the candidate buffer
    for (a0, a1) in ep[1:]:
        assert a1 <= a0 * ops.KNN_EPOCH_GROWTH or a1 == gy     # thresholds are never staler than one growth step


def test_north_star_point_is_one_cold_round_and_one_warm_epoch():
    assert ops.knn_epochs(1500, 30000, 256) == [(0, 42), (42, 118)]       # 6 x 42 = 252 workgroups, then 6 x 76 = 456
