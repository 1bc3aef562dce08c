"""Host logic of the three-product GEMM's launch (csrc/gemm_split2.hip: x2_tile_height, x2_xcd_tiles) without a GPU: the kernel's schedule
hands every XCD (blockIdx & 7) a contiguous range of M-panels, so a launch costs the ROUNDS of its busiest XCD (256 CUs: 32 workgroups each)
times the height of a tile over that height's efficiency; the launcher picks 256 / 192 / 128 rows by that count and starts one workgroup
per tile of the busiest XCD.  The expectations are the measured winners of tools/x2_cut_probe.py (profiles/r06_x2_cut_probe.log)."""
import pytest

from fitclip_amd import ops

CUS = 256


def _xcd_tiles(M, N, rows, groups=1):
    panels, cols = -(-M // rows), -(-N // 256)
    per = 8 // groups
    return -(-panels // per) * (cols // groups)


@pytest.mark.parametrize("M,N,want", [
    (403456, 3072, 256), (403456, 768, 256), (403456, 2304, 256),          # the bench pass: the big tile
    (25216, 768, 192), (25216, 2304, 256), (25216, 3072, 256),             # the reference's eval batch (128 frames): out_proj / c_proj 297 big tiles
    (12608, 768, 192), (12608, 2304, 256),
    (6304, 768, 128), (6304, 2304, 128), (6304, 3072, 192),                # a 32-frame call
    (2464, 512, 128), (2464, 1536, 128), (2464, 2048, 128),                # the text tower of 32 captions
    (19712, 512, 192), (19712, 1536, 256), (9856, 2048, 192), (9856, 512, 128), (9856, 1536, 256),
    (50432, 768, 256), (100864, 768, 256), (1, 32, 128),
])
def test_tile_height_by_rounds_of_the_busiest_xcd(M, N, want):
    rows, wgs = ops.gemm_split2_plan(M, N, CUS)
    assert rows == want, (M, N, rows)
    groups = 4 if (N // 256) % 4 == 0 and N // 256 >= 8 and N % 256 == 0 else 1   # (wide problems: the N range over 4 XCD groups)
    assert wgs == min(8 * _xcd_tiles(M, N, rows, groups), CUS)


def test_every_tile_of_the_busiest_xcd_has_a_workgroup():
    """The case that cost a second round before: 20 panels of 128 rows x 2 column tiles - 6 tiles on XCDs 0-3, 4 on the others; min(tiles,
    CUs) = 40 workgroups were 5 per XCD."""
    rows, wgs = ops.gemm_split2_plan(2464, 512, CUS)
    assert (rows, wgs) == (128, 48)
    for M in range(1, 70000, 977):
        for N in (32, 512, 768, 1536, 2048, 2304, 3072, 4096):
            rows, wgs = ops.gemm_split2_plan(M, N, CUS)
            assert rows in (256, 192, 128) and 8 <= wgs <= CUS and wgs % 8 == 0
            groups = 4 if (N // 256) % 4 == 0 and N // 256 >= 8 and N % 256 == 0 else 1
            assert wgs == min(8 * _xcd_tiles(M, N, rows, groups), CUS)


def test_plan_rejects_bad_arguments():
    from fitclip_amd import _lib
    with pytest.raises(_lib.FitclipHipError):
        ops.gemm_split2_plan(0, 768, CUS)
