"""CPU check of the data movement of the experimental 256x256 tile GEMM (csrc/gemm.hip, gemm_tile4_kernel): emulates the per-lane
global -> LDS staging (global_load_lds_dwordx4: lane l of a wave lands at M0 base + 16 l) and the per-lane fragment reads with the
kernel's own index formulas, and verifies that every MFMA operand fragment holds A[row][k0 + 8 lg .. + 7] of the row the
accumulator layout expects, and that the 16 lanes of a fragment-read group hit 64 distinct LDS banks.  No GPU needed.

    python tools/proto/tile4_addressing_check.py
"""
import numpy as np

T4_M, ROWB, A_BYTES = 256, 64, 256 * 64


def stage_tile(tile, nth=256):
    """tile: [256][32] element ids of one operand tile for one K slab -> LDS image (16384 bytes worth of element ids, 2 bytes each).
    nth = 256 (4 waves, 4 chunks per thread) or 512 (8 waves, 2 chunks per thread)."""
    lds = np.full(A_BYTES // 2, -1, dtype=np.int64)
    for tid in range(nth):
        wave, lane = tid >> 6, tid & 63
        for i in range(1024 // nth):
            c = tid + i * nth
            row, slot = c >> 2, c & 3
            lch = slot ^ ((row >> 2) & 3)                       # source chunk of this LDS slot
            byte = wave * 1024 + i * (nth * 16) + lane * 16     # M0 base (wave_base + i * NTH * 16) + 16 * lane
            assert byte == c * 16
            lds[byte // 2: byte // 2 + 8] = tile[row, lch * 8: lch * 8 + 8]
    assert (lds >= 0).all()
    return lds


def frag(lds, wm, i, lane):
    l15, lg = lane & 15, lane >> 4
    ra = wm + i * 16 + l15
    off = ra * ROWB + ((lg ^ ((ra >> 2) & 3)) << 4)
    return ra, lg, lds[off // 2: off // 2 + 8], off


def main():
    tile = np.arange(256 * 32).reshape(256, 32)                 # element id = row * 32 + k
    lds = stage_tile(tile)
    assert (stage_tile(tile, 512) == lds).all()                 # the 8-wave variant fills the same image
    for wm in (0, 128):
        for i in range(8):
            banks_per_group = {}
            for lane in range(64):
                ra, lg, vals, off = frag(lds, wm, i, lane)
                assert (vals == tile[ra, lg * 8: lg * 8 + 8]).all(), (wm, i, lane)
                # ds_read_b128 is served 16 lanes at a time: those lanes' 16-byte pieces must cover 64 distinct 4-byte banks
                banks_per_group.setdefault(lane >> 4, set()).update(((off // 4) + w) % 64 for w in range(4))
            assert all(len(b) == 64 for b in banks_per_group.values()), banks_per_group
    # epilogue staging pitch: fragment writes of one pass (rows lg*4 + r, columns j*16 + l15) are conflict-free per 16-lane group... and
    # the 64 lanes of a write instruction (fixed j, r) hit 64 distinct banks
    for PITCH, wnf in ((132, 8), (68, 4)):
        for j in range(wnf):
            for r in range(4):
                banks = {((lg * 4 + r) * PITCH + j * 16 + l15) % 64 for lg in range(4) for l15 in range(16)}
                assert len(banks) == 64
    print("tile4 addressing: staging, fragment reads and bank mapping OK")


if __name__ == "__main__":
    main()
