"""CPU: the LDS layout of attention128.hip (K / Vt tiles written by LDS-DMA with a source-side XOR swizzle, read back as 16-byte MFMA
fragments) as a numpy-free model: every fragment read returns the chunk the MFMA expects, and the 16 lanes the hardware services together
(`ds_read_b128` lane groups of MI355X_MICROARCH.md's LDS table) hit 16 different 16-byte columns - no bank conflicts.  This is the check the
kernel's rewrite was run against before its first GPU test."""

# lane groups of one ds_read_b128 (lower half of the wave; the upper half is the same + 32)
GROUPS = [[0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27], [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31]]


def kperm(li):      # attention128.hip: K rows in swap_bits23 order
    return (li & 19) | ((li & 4) << 1) | ((li & 8) >> 1)


def dma_image(tile_bytes, row_bytes, src_chunk):
    """LDS byte offset -> (row, logical chunk) as the DMAs of one stage leave it: block (i * 4 + wave) of 1 KB, lane * 16 inside"""
    lds = {}
    for i in range(tile_bytes // 4096):
        for wave in range(4):
            for lane in range(64):
                o = (i * 4 + wave) * 1024 + lane * 16
                r, pch = o // row_bytes, (o % row_bytes) >> 4
                lds[o] = (r, src_chunk(r, pch))
    assert len(lds) == tile_bytes // 16 and len(set(lds.values())) == tile_bytes // 16        # a bijection
    return lds


def test_k_tile_fragments_and_banks():
    for parts in (1, 2):                                    # hi only / hi + lo (SPLIT)
        krow = parts * 256
        lds = dma_image(32 * krow, krow, lambda r, p: (p & ~15) | ((p ^ r) & 15))
        for ks in range(8):
            for lh in range(2):
                addr = {}
                for li in range(32):
                    row = kperm(li)
                    a = row * krow + (((2 * ks + lh) ^ row) & 15) * 16
                    assert lds[a] == (row, 2 * ks + lh)
                    if parts == 2:
                        assert lds[a + 256] == (row, 16 + 2 * ks + lh)
                    addr[li] = a
                for g in GROUPS:
                    assert len({(addr[li] % 256) // 16 for li in g}) == 16


def test_vt_tile_fragments_and_banks():
    for vr in (32, 64, 128, 256):                           # Vt rows per tile: NVB * 32, twice that with P V split
        vrs = max(vr, 64)
        lds = dma_image(vrs * 64, 64, lambda r, p: p ^ ((r >> 2) & 3))
        lds = {o: ((r if vrs == vr else r % vr), c) for o, (r, c) in lds.items()}
        for row0 in range(0, vr, 32):
            for s in range(2):
                for lh in range(2):
                    addr = {}
                    for li in range(32):
                        row = row0 + li
                        a = row * 64 + ((2 * s + lh) ^ ((li >> 2) & 3)) * 16
                        assert lds[a] == (row, 2 * s + lh)
                        addr[li] = a
                    for g in GROUPS:
                        assert len({(addr[li] % 256) // 16 for li in g}) == 16
