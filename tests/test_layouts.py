"""LDS layout algebra of the kernels, restated in Python (CPU): the XOR swizzles applied to the LDS-DMA source
addresses must be (1) permutations inside a row and (2) conflict-free for the read instruction's lane groups, using
the bank model of /opt/skills/guides/MI355X_MICROARCH.md §LDS: 64 banks x 4 B (256 B per clock) for ds_read_b128 /
ds_read_b64_tr_b16, lane groups of one LDS cycle each:
    ds_read_b128        {0-3,12-15,20-27} {4-11,16-19,28-31} {32-35,44-47,52-59} {36-43,48-51,60-63}
    ds_read_b64_tr_b16  {0-31} {32-63}
A group is conflict-free when its lanes touch pairwise different banks (or identical addresses).
These are the formulas of hgemm_pingpong.hip / hgemm_w4.hip (st_2x8, 128-B rows) and of
attn_fwd.hip / attn_w4u.hip / attn_bigd2.hip (K: chunk ^ (row & 15) on 256-B rows; V: 64-B unit ^ (row & 3) for the transpose reads)."""
import itertools

B128_GROUPS = [
    [0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27],
    [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31],
    [32, 33, 34, 35, 44, 45, 46, 47, 52, 53, 54, 55, 56, 57, 58, 59],
    [36, 37, 38, 39, 40, 41, 42, 43, 48, 49, 50, 51, 60, 61, 62, 63],
]
TR_GROUPS = [list(range(0, 32)), list(range(32, 64))]


def banks(addr, nbytes):
    return {(addr + b) // 4 % 64 for b in range(0, nbytes, 4)}


def conflict_free(addrs, nbytes):
    seen = {}
    for a in addrs:
        for bk in banks(a, nbytes):
            if bk in seen and seen[bk] != a:
                return False
            seen[bk] = a
    return True


def test_gemm_st2x8_swizzle_is_conflict_free_for_b128_fragment_reads():
    # 128-B rows; 16-B chunk c of row r at slot c ^ ((r >> 1) & 7); lane -> row l32 (+32 per block), chunk 2ks + hi
    for ks in range(4):
        for grp in B128_GROUPS:
            addrs = []
            for lane in grp:
                l32, hi = lane & 31, lane >> 5
                addrs.append(l32 * 128 + (((2 * ks + hi) ^ ((l32 >> 1) & 7)) * 16))
            assert conflict_free(addrs, 16), (ks, grp)
    for r in range(16):   # a permutation of the 8 chunk slots in every row
        assert sorted(c ^ ((r >> 1) & 7) for c in range(8)) == list(range(8))


def test_gemm_st2x8_swizzle_is_conflict_free_for_16x16x32_fragment_reads():
    # hgemm_w4x.hip: the SAME image (128-B rows, chunk c of row r at slot c ^ ((r >> 1) & 7)) read for v_mfma_f32_16x16x32:
    # lane -> row 16 i + (l & 15), chunk 4 ks + (l >> 4)
    for ks, i in itertools.product(range(2), range(8)):
        for grp in B128_GROUPS:
            addrs = []
            for lane in grp:
                row, kg = 16 * i + (lane & 15), lane >> 4
                addrs.append(row * 128 + (((4 * ks + kg) ^ ((row >> 1) & 7)) * 16))
            assert conflict_free(addrs, 16), (ks, i, grp)
            # the kernel's address form: (l >> 1) & 7 == (row >> 1) & 7 for row = 16 i + (l & 15)
            for lane in grp:
                assert ((lane >> 1) & 7) == (((16 * i + (lane & 15)) >> 1) & 7)


def test_attention_k_tile_swizzle_256_byte_rows():
    # K tile [64 kv][256 B]: chunk c of row r at slot c ^ (r & 15); fragment lane -> row tt*32 + l32, chunk 2ks + hi
    for ks, tt in itertools.product(range(8), range(2)):
        for grp in B128_GROUPS:
            addrs = []
            for lane in grp:
                l32, hi = lane & 31, lane >> 5
                row = tt * 32 + l32
                addrs.append(row * 256 + (((2 * ks + hi) ^ (row & 15)) * 16))
            assert conflict_free(addrs, 16), (ks, tt, grp)


def test_attention_v_tile_unit_swizzle_for_transpose_reads():
    # V tile [64 kv][256 B]: 64-B unit u of row r at unit u ^ (r & 3).  A transpose read: lane i of a 16-lane group
    # supplies row (i >> 2), 8 bytes at column 4 (i & 3) of a 16-column block; lanes 16..31 take the next 16 columns;
    # lane half hi takes rows +4.
    for g, x, dt in itertools.product(range(4), range(2), range(4)):
        for grp in TR_GROUPS:
            addrs = []
            for lane in grp:
                i, gi, hi = lane & 15, (lane >> 4) & 1, lane >> 5
                row = 32 * (g >> 1) + 16 * (g & 1) + 8 * x + 4 * hi + (i >> 2)
                unit = dt ^ (row & 3)
                addrs.append(row * 256 + unit * 64 + 32 * gi + 8 * (i & 3))
            assert conflict_free(addrs, 8), (g, x, dt)


def test_nn_b_image_pair_swizzle_for_transpose_reads():
    # NN B sub-image [k][256 B] = 128 contiguous columns; 32-B pair P of row k at pair slot P ^ ((k & 3) << 1)
    for j, ks, x in itertools.product(range(4), range(4), range(2)):
        for grp in TR_GROUPS:
            addrs = []
            for lane in grp:
                i, gi, hi = lane & 15, (lane >> 4) & 1, lane >> 5
                k = 16 * ks + 8 * hi + (i >> 2) + 4 * x
                pair = (2 * j + gi) ^ ((k & 3) << 1)
                addrs.append(k * 256 + pair * 32 + (i & 3) * 8)
            assert conflict_free(addrs, 8), (j, ks, x)
    for k in range(4):
        assert sorted(p ^ ((k & 3) << 1) for p in range(8)) == list(range(8))


def test_attention_tiles_for_16x16x32_fragments():
    # attn_w4u.hip (D = 128).  K tile (unchanged image: chunk c of row r at slot c ^ (r & 15), 256-B rows): fragment lane -> row
    # 16 kvb + (l & 15), chunk 4 ds + (l >> 4)
    for ds, kvb in itertools.product(range(4), range(4)):
        for grp in B128_GROUPS:
            addrs = [(16 * kvb + (l & 15)) * 256 + (((4 * ds + (l >> 4)) ^ (l & 15)) * 16) for l in grp]
            assert conflict_free(addrs, 16), (ds, kvb)
    # V tile: 32-B pair p of row r at pair slot p ^ key(r), key(r) = ((r & 3) << 1) | ((r >> 2) & 1); a transpose read: lane i
    # of 16-lane group g supplies row 32 H + 16 x + 4 g + (i >> 2), 8 bytes at column 4 (i & 3) of pair db
    def key(r):
        return ((r & 3) << 1) | ((r >> 2) & 1)
    for db, hh, x in itertools.product(range(8), range(2), range(2)):
        for grp in TR_GROUPS:
            addrs = []
            for lane in grp:
                i, g = lane & 15, lane >> 4
                r = 32 * hh + 16 * x + 4 * g + (i >> 2)
                assert key(r) == (((i >> 2) << 1) | (g & 1))
                a = r * 256 + (db ^ key(r)) * 32 + (i & 3) * 8
                # the kernel's address form: pair 2u at (2u ^ key) * 32, pair 2u + 1 at +32 (g even) / -32 (g odd)
                u = db >> 1
                a2 = r * 256 + ((2 * u) ^ key(r)) * 32 + (i & 3) * 8 + ((db & 1) * (-32 if (g & 1) else 32))
                assert a == a2
                addrs.append(a)
            assert conflict_free(addrs, 8), (db, hh, x)
        # (round 2.s 64-B-unit swizzle u ^ (r & 3) is 2-way conflicted for this read pattern)
        bad = []
        for lane in TR_GROUPS[0]:
            i, g = lane & 15, lane >> 4
            r = 32 * hh + 16 * x + 4 * g + (i >> 2)
            bad.append(r * 256 + ((db >> 1) ^ (r & 3)) * 64 + (db & 1) * 32 + (i & 3) * 8)
        assert not conflict_free(bad, 8)
    # DMA side: the lane filling 16-B slot cs of row r = 4 p + r4 fetches logical chunk ((cs >> 1) ^ key(r)) * 2 + (cs & 1)
    for p, r4, cs in itertools.product(range(16), range(4), range(16)):
        r = 4 * p + r4
        assert key(r) == ((r4 << 1) | (p & 1))
        chunk = (((cs >> 1) ^ key(r)) << 1) | (cs & 1)
        assert ((chunk >> 1) ^ key(r)) == (cs >> 1) and (chunk & 1) == (cs & 1)     # reading pair P finds it at slot P ^ key


def test_nn_b_image_key_for_16x16x32_transpose_reads():
    # hgemm_w4y.hip NN: 32-B pair P of row k at pair slot P ^ key(k), key(k) = ((k & 3) << 1) | ((k >> 3) & 1); a B fragment
    # read: lane i of 16-lane group g supplies k row 32 ks + 8 g + 4 x + (i >> 2), 8 bytes at column 4 (i & 3) of pair j
    def key(k):
        return ((k & 3) << 1) | ((k >> 3) & 1)
    for j, ks, x in itertools.product(range(8), range(2), range(2)):
        for grp in TR_GROUPS:
            addrs = []
            for lane in grp:
                i, g = lane & 15, lane >> 4
                k = 32 * ks + 8 * g + 4 * x + (i >> 2)
                assert key(k) == (((i >> 2) << 1) | (g & 1))          # the kernel's lane-constant form
                addrs.append(k * 256 + (j ^ key(k)) * 32 + (i & 3) * 8)
            assert conflict_free(addrs, 8), (j, ks, x)
        # (the 32x32x16 kernel's key (k & 3) << 1 alone would put rows k and k + 8 of a 32-lane group on the same banks)
        bad = [(32 * ks + 8 * (lane >> 4) + 4 * x + ((lane & 15) >> 2)) * 256 + (j ^ (((lane & 15) >> 2) << 1)) * 32
               + (lane & 3) * 8 for lane in TR_GROUPS[0]]
        assert not conflict_free(bad, 8)
    for k in range(64):   # a permutation of the 8 pair slots in every row; DMA side: (k >> 3) & 1 == (p2 >> 1) & 1, k = 4 (4 w + p2) + r
        assert sorted(p ^ key(k) for p in range(8)) == list(range(8))
    for w, p2, r in itertools.product(range(4), range(4), range(4)):
        k = 4 * (4 * w + p2) + r
        assert key(k) == ((r << 1) | ((p2 >> 1) & 1))


def test_dma_source_permutation_is_the_inverse_of_the_read_mapping():
    # LDS-DMA writes lane-linearly: the lane that fills slot s of row r must FETCH logical chunk s ^ key(r); reading
    # logical chunk c then finds it at slot c ^ key(r) (XOR is an involution) — for every key used above
    for key in (lambda r: (r >> 1) & 7, lambda r: r & 15, lambda r: (r >> 2) & 3):
        width = 16 if key(15) == 15 else (8 if key(15) == 7 else 4)
        for r in range(64):
            image = {s: s ^ key(r) for s in range(width)}          # slot -> logical chunk stored there
            for c in range(width):
                assert image[c ^ key(r)] == c


def test_bigd2_tiles_on_long_rows():
    """attn_bigd2.hip (D = 256 / 512: 512-B / 1-KiB rows).  K: 16-B chunk c of row r at slot (c & ~15) | ((c ^ r) & 15);
    fragment lane -> row tt*32 + l32, chunk 2 ks + hi.  V: 64-B unit u of row r at unit u ^ (r & 3) inside its 256-B group;
    transpose read of step (g, dq), d tile j: lane -> kv row 16 g + 8 x + 4 hi + (i >> 2), 32-column tile dt = 4 dq + j.
    Parked Q: lane-private 16-B slots, 1 KiB per k-step."""
    for D in (256, 512):
        rowb = 2 * D
        for ks, tt in itertools.product(range(D // 16), range(2)):
            for grp in B128_GROUPS:
                addrs = []
                for lane in grp:
                    l32, hi = lane & 31, lane >> 5
                    row = tt * 32 + l32
                    # the kernel's form: kx[ks & 7] + (ks >> 3) * 256, kx = l32 * ROWB + (((2 k8 + hi) ^ (l32 & 15)) * 16)
                    addrs.append(row * rowb + (((2 * (ks & 7) + hi) ^ (l32 & 15)) * 16) + (ks >> 3) * 256)
                    c = 2 * ks + hi                      # == the layout's definition
                    assert addrs[-1] == row * rowb + ((c & ~15) | ((c ^ row) & 15)) * 16
                assert conflict_free(addrs, 16), (D, ks, tt, grp)
        ndt = D // 32
        for g, x, dt in itertools.product(range(4), range(2), range(ndt)):
            for grp in TR_GROUPS:
                addrs = []
                for lane in grp:
                    i, gi, hi = lane & 15, (lane >> 4) & 1, lane >> 5
                    # kernel: vx[j] + dq * 256 + g * 16 * ROWB (+ 8 * ROWB), vx[j] = (4 hi + (i >> 2)) * ROWB + 32 gi + 8 (i & 3)
                    #         + ((j ^ (i >> 2)) << 6)
                    j, dq = dt & 3, dt >> 2
                    a = (4 * hi + (i >> 2)) * rowb + 32 * gi + 8 * (i & 3) + ((j ^ (i >> 2)) << 6) + dq * 256 \
                        + g * 16 * rowb + x * 8 * rowb
                    row = 16 * g + 8 * x + 4 * hi + (i >> 2)
                    unit = (dt & ~3) | ((dt ^ row) & 3)
                    assert a == row * rowb + unit * 64 + 32 * gi + 8 * (i & 3)
                    addrs.append(a)
                assert conflict_free(addrs, 8), (D, g, x, dt)
        # LDS-DMA source side: lane slot cs of a 1-KiB piece receives source chunk cs ^ key -> a permutation of the row's chunks
        cpr = rowb // 16
        for row in range(64):
            assert sorted((cs & ~15) | ((cs ^ row) & 15) for cs in range(cpr)) == list(range(cpr))          # K
            assert sorted(cs ^ ((row & 3) << 2) for cs in range(cpr)) == list(range(cpr))                   # V
    for grp in B128_GROUPS:                                  # parked Q fragments
        assert conflict_free([lane * 16 for lane in grp], 16)


def test_xcd_super_block_raster_is_a_bijection_with_compact_steps():
    """hgemm_mfma256.hip raster_xcd16 restated: block b (XCD b % 8, slot b >> 3) -> C tile.  (1) a bijection onto the tile
    grid for every grid incl. ragged ones (49 x 49 = 12544 / 256, 60 x 60, rectangular); (2) on a grid of full 16 x 16 blocks
    every step of 256 consecutive blocks is ONE 16 x 16 block of tiles and every XCD's 32 tiles of the step a 4 x 8 block."""
    def tile(b, nwg, tm_, tn_):
        full = nwg & ~255
        i = b
        if b < full:
            x, idx = b & 7, b >> 3
            i = ((idx >> 5) << 8) + (x << 5) + (idx & 31)
        per_panel = 16 * tm_
        panel, rem = divmod(i, per_panel)
        pn0 = panel * 16
        w = min(16, tn_ - pn0)
        if w == 16:
            grp, r2 = rem >> 8, rem & 255
            if 16 * grp + 16 <= tm_:
                sub, cc = r2 >> 5, r2 & 31
                return 16 * grp + 4 * (sub >> 1) + (cc >> 3), pn0 + 8 * (sub & 1) + (cc & 7)
            return 16 * grp + (r2 >> 4), pn0 + (r2 & 15)
        return rem // w, pn0 + rem % w
    for tm_, tn_ in ((32, 32), (64, 64), (49, 49), (60, 60), (61, 63), (16, 48), (13, 7), (3, 100), (24, 40)):
        nwg = tm_ * tn_
        seen = {tile(b, nwg, tm_, tn_) for b in range(nwg)}
        assert len(seen) == nwg and all(0 <= a < tm_ and 0 <= c < tn_ for a, c in seen), (tm_, tn_)
    tm_ = tn_ = 64
    nwg = tm_ * tn_
    for step in range(nwg // 256):
        tiles = [tile(b, nwg, tm_, tn_) for b in range(256 * step, 256 * step + 256)]
        rows, cols = {a for a, _ in tiles}, {c for _, c in tiles}
        assert len(rows) == 16 and len(cols) == 16 and max(rows) - min(rows) == 15 and max(cols) - min(cols) == 15
        for x in range(8):
            mine = [tile(b, nwg, tm_, tn_) for b in range(256 * step, 256 * step + 256) if b % 8 == x]
            assert len({a for a, _ in mine}) == 4 and len({c for _, c in mine}) == 8


def test_attention_tiles_d64_for_16x16x32_fragments():
    """attn_w4u.hip at D = 64 (128-B rows: two rows per 256-B bank row).  K: 16-B chunk c of row r at slot c ^ ((r >> 1) & 7);
    fragment lane -> row 16 kvb + (l & 15), chunk 4 ds + (l >> 4).  V: 32-B pair p of row r at pair slot p ^ ((r >> 1) & 3);
    transpose read: lane i of 16-lane group g supplies row 32 H + 16 x + 4 g + (i >> 2), 8 bytes at column 4 (i & 3) of pair
    db.  DMA: a 1-KiB piece = 8 rows; wave w stages pieces w, w + 4, so row & 15 = 8 (w & 1) + rr for both of them."""
    for ds, kvb in itertools.product(range(2), range(4)):
        for grp in B128_GROUPS:
            addrs = []
            for l in grp:
                row = 16 * kvb + (l & 15)
                a = row * 128 + (((4 * ds + (l >> 4)) ^ ((row >> 1) & 7)) * 16)
                assert a == (16 * kvb) * 128 + (l & 15) * 128 + (((4 * ds + (l >> 4)) ^ (((l & 15) >> 1) & 7)) * 16)   # kernel form
                addrs.append(a)
            assert conflict_free(addrs, 16), (ds, kvb)
    def key(r):
        return (r >> 1) & 3
    for db, hh, x in itertools.product(range(4), range(2), range(2)):
        for grp in TR_GROUPS:
            addrs = []
            for lane in grp:
                i, g = lane & 15, lane >> 4
                r = 32 * hh + 16 * x + 4 * g + (i >> 2)
                assert key(r) == (((g & 1) << 1) | (i >> 3))            # the kernel's lane-constant form
                addrs.append(r * 128 + (db ^ key(r)) * 32 + (i & 3) * 8)
            assert conflict_free(addrs, 8), (db, hh, x)
    for r in range(64):
        assert sorted(c ^ ((r >> 1) & 7) for c in range(8)) == list(range(8))
        assert sorted(p ^ key(r) for p in range(4)) == list(range(4))
    for w, i2, rr, cs in itertools.product(range(4), range(2), range(8), range(8)):
        r = 8 * (w + 4 * i2) + rr
        assert ((r >> 1) & 7) == 4 * (w & 1) + (rr >> 1) and key(r) == ((rr >> 1) & 3)
        chunk = cs ^ ((r >> 1) & 7)                                   # the lane filling slot cs fetches this logical chunk
        assert chunk ^ ((r >> 1) & 7) == cs
        vchunk = (((cs >> 1) ^ key(r)) << 1) | (cs & 1)
        assert ((vchunk >> 1) ^ key(r)) == (cs >> 1) and (vchunk & 1) == (cs & 1)


B64_GROUPS = [list(range(0, 32)), list(range(32, 64))]     # ds_read_b64: 2 x 32 lanes, bank (a / 4) mod 64 (MI355X_MICROARCH.md LDS table)


def test_attention_v_transposed_image_for_plain_b64_fragment_reads():
    """attn_w4u.hip, VT: V handed over as [D][N] (the reference's *_swizzle_qkv entries).  A tile's image is [D rows][64 kv] —
    128-B rows whatever D is (two rows per 256-B bank row) — with 16-B granule j of row d at slot j ^ ((d >> 1) & 7).
    The Vᵀ operand of a v_mfma_f32_16x16x32 block db, lane (l16, g4): d-row 16 db + l16, k slots 8 g4 + e <-> kv = 32 H +
    16 (e >> 2) + 4 g4 + (e & 3) (the order the lane-local Pᵀ operand defines) = TWO plain ds_read_b64: half (g4 & 1) of granule
    4 H + 2 x + (g4 >> 1), x = 0 / 1.  (1) conflict-free per lane group, (2) the kernel's address form, (3) the values a lane gets are
    the right kv columns, (4) the DMA source permutation is the inverse: a piece = 8 d-rows, wave w stages pieces w, w + 4, ..."""
    for D in (64, 128):
        for db, H, x in itertools.product(range(D // 16), range(2), range(2)):
            for grp in B64_GROUPS:
                addrs = []
                for lane in grp:
                    l16, g4 = lane & 15, lane >> 4
                    d = 16 * db + l16
                    gran = 4 * H + 2 * x + (g4 >> 1)
                    a = d * 128 + ((gran ^ ((d >> 1) & 7)) * 16) + 8 * (g4 & 1)
                    # kernel form: vx[2 H + x] + db * 2048 with vx[u] = l16 * 128 + (((2 u) | (g4 >> 1)) ^ ((l16 >> 1) & 7)) * 16 + 8 (g4 & 1)
                    u = 2 * H + x
                    assert a == db * 2048 + l16 * 128 + ((((2 * u) | (g4 >> 1)) ^ ((l16 >> 1) & 7)) * 16) + 8 * (g4 & 1)
                    addrs.append(a)
                    # which kv columns sit there: slot s of row d holds granule s ^ key(d), i.e. kv 8 (s ^ key) .. + 7
                    slot = (a % 128) // 16
                    kv0 = 8 * (slot ^ ((d >> 1) & 7)) + 4 * ((a % 16) // 8)
                    assert kv0 == 32 * H + 16 * x + 4 * g4
                assert conflict_free(addrs, 8), (D, db, H, x)
        for r in range(D):
            assert sorted(j ^ ((r >> 1) & 7) for j in range(8)) == list(range(8))
        # DMA: piece p = d-rows 8 p .. 8 p + 7; lane -> rr = lane >> 3, LDS granule slot cs = lane & 7 <- source granule cs ^ key(row)
        for w, i, rr, cs in itertools.product(range(4), range(D // 32), range(8), range(8)):
            p = w + 4 * i
            row = 8 * p + rr
            assert ((row >> 1) & 7) == 4 * (w & 1) + (rr >> 1)        # the kernel's lane-constant key (p & 1 == w & 1)
            src = cs ^ (4 * (w & 1) + (rr >> 1))
            assert src ^ ((row >> 1) & 7) == cs
        # every d-row of the tile is staged exactly once
        rows = sorted(8 * (w + 4 * i) + rr for w in range(4) for i in range(D // 32) for rr in range(8))
        assert rows == list(range(D))


def test_hgemm_w4y_piece_map_covers_every_row_once_and_keeps_the_image():
    """hgemm_w4y.hip / gemm_fp8.hip piece map (round 3): piece g of wave w stages rows 32 g + 8 w .. + 8 of the 256-row K-contiguous
    tile — the four waves' concurrent requests are 32 CONSECUTIVE rows — into LDS bytes g * 4096 + w * 1024 .. + 1024 of the slot.
    (1) every row of the tile is staged exactly once and lands at row * 128 (the row-major image the fragment reads assume);
    (2) the source chunk a lane fetches is the one its LDS slot must hold under the st_2x8 swizzle (slot c of row r holds chunk
        c ^ ((r >> 1) & 7)), with ONE lane-offset register per wave: the key's bit 2 is (8-row block & 1) = wave & 1 for every piece;
    (3) at one piece index g the four waves cover a contiguous 32-row span."""
    seen = {}
    for wave in range(4):
        for g in range(8):
            first = 32 * g + 8 * wave
            lds0 = g * 4096 + wave * 1024
            for lane in range(64):
                row = first + (lane >> 3)
                slot = lane & 7
                src_chunk = slot ^ (((lane >> 4) & 3) | ((wave & 1) << 2))        # a_off of the kernel (par = wave & 1)
                lds = lds0 + lane * 16                                          # LDS-DMA writes lane-linearly
                assert lds == row * 128 + slot * 16
                assert src_chunk == slot ^ ((row >> 1) & 7)
                assert (row, slot) not in seen
                seen[(row, slot)] = src_chunk
    assert len(seen) == 256 * 8
    for row in range(256):
        assert sorted(seen[(row, s)] for s in range(8)) == list(range(8))     # a permutation inside the row
    for g in range(8):
        rows = sorted(32 * g + 8 * w + i for w in range(4) for i in range(8))
        assert rows == list(range(32 * g, 32 * g + 32))


def test_k_loop_stagger_index_sequence_visits_every_tile_once():
    """The generated K loop of hgemm_w4y (tools/gen_hgemm_w4y.py STAGGER) maps the logical tile index i — clamped to KT - 1 for the
    prefetches past the end — to memory tile (i + stg) mod KT with stg < KT, using add / compare / conditional subtract.  Over a
    whole walk every memory tile is the operand of exactly one logical tile, for every KT and stagger the launcher can produce
    (auto: stg = XCD * max(1, KT / 8) mod KT)."""
    def mem_tile(i, kt, stg):
        x = min(i, kt - 1) + stg
        return x - kt if x >= kt else x

    for kt in list(range(1, 20)) + [32, 64, 128, 196, 256]:
        step = max(1, kt // 8)
        for xcd in range(8):
            stg = (xcd & 7) * step % kt
            assert 0 <= stg < kt
            walk = [mem_tile(t, kt, stg) for t in range(kt)]
            assert sorted(walk) == list(range(kt))
            # the loop prefetches logical tiles t + 2 (clamped): always a valid tile of the matrix
            assert all(0 <= mem_tile(t + 2, kt, stg) < kt for t in range(kt))


def test_persistent_attention_walk_visits_every_block_once_on_its_xcd():
    """attn_w4u.hip WALK = 1: workgroup w of a G-workgroup grid (G = min(blocks, CUs)) walks the virtual block ids w, w + G, w + 2 G, ... and maps
    each through xcd_remap(v, nblk) (lc_common.h: virtual id v lives on XCD v & 7; an XCD owns a contiguous range of real ids).  Every
    real block is computed exactly once, and — G being a multiple of 8 whenever there is more than one round — all blocks of a
    workgroup stay on ITS XCD, whose L2 holds their heads' K / V (the GPU tests run 296 / 516 / 48 / 260 blocks)."""
    def xcd_remap(b, nwg):
        q, r, xcd, idx = nwg >> 3, nwg & 7, b & 7, b >> 3
        base = xcd * (q + 1) if xcd < r else r * (q + 1) + (xcd - r) * q
        return base + idx

    for ncu in (256, 64, 8):
        for nblk in (1, 5, 48, 255, 256, 257, 260, 296, 516, 2048, 2051):
            G = min(nblk, ncu)
            seen = []
            for w in range(G):
                v = w
                while v < nblk:
                    seen.append(xcd_remap(v, nblk))
                    if nblk > G:
                        assert v & 7 == w & 7        # G % 8 == 0 here: the walk stays on the workgroup's XCD
                    v += G
            assert sorted(seen) == list(range(nblk)), (ncu, nblk)


def test_dynamic_attention_walk_claims_every_block_once_on_its_xcd():
    """attn_w4u.hip WALK = 2: workgroup w starts on virtual block w; afterwards a workgroup on XCD x = w & 7 claims j = the next value
    of XCD x's counter and runs virtual block G + x + 8 j (ids >= G with id & 7 == x, in claim order) until the id reaches nblk.
    Whatever the interleaving of the claims, every real block is computed exactly once, a workgroup never leaves its XCD, and an
    XCD's blocks are handed out in ascending real-id order (= by head: the L2 locality the one-block launch gets from the hardware
    dispatcher).  The counters end at one failed claim per workgroup: what the last workgroup out resets."""
    import random

    def xcd_remap(b, nwg):
        q, r, xcd, idx = nwg >> 3, nwg & 7, b & 7, b >> 3
        base = xcd * (q + 1) if xcd < r else r * (q + 1) + (xcd - r) * q
        return base + idx

    rng = random.Random(4)
    for G, nblk in ((256, 257), (256, 260), (256, 516), (256, 4096), (64, 1000), (8, 77)):
        counters = [0] * 8
        seen, per_xcd = [], {x: [] for x in range(8)}
        live = list(range(G))
        for w in live:
            seen.append(xcd_remap(w, nblk))
            per_xcd[w & 7].append(xcd_remap(w, nblk))
        while live:                                   # workgroups finish their blocks in random order
            w = live.pop(rng.randrange(len(live)))
            x = w & 7
            j = counters[x]
            counters[x] += 1
            v = G + x + 8 * j
            if v < nblk:
                assert v & 7 == x
                seen.append(xcd_remap(v, nblk))
                per_xcd[x].append(xcd_remap(v, nblk))
                live.append(w)                        # it will claim again after this block
        assert sorted(seen) == list(range(nblk)), (G, nblk)
        for x in range(8):
            tail = per_xcd[x][G // 8:]
            assert tail == sorted(tail)               # claim order = ascending real ids inside the XCD's contiguous range
        assert sum(counters) == (nblk - G) + G        # every successful claim + exactly one failed claim per workgroup


def test_hgemm_w4y_epilogue_staging_fits_one_ring_slot_and_is_conflict_free():
    """hgemm_w4y.hip w4y_epilogue_b2 (round 4: the cross-tile prefetch owns A slots 0 / 1 and B slots 0 / 1, so the C tile leaves through B
    slot 2 alone): four waves x 32 rows x 256 B = exactly 32 KiB, chunk c of row r at chunk c ^ (r & 15).  Written with 8-byte stores
    (lane (r16, kg): row 16 ih + r16, logical chunk 2 j + (kg >> 1), half kg & 1), read back as whole rows with ds_read_b128 (lane ->
    row 4 it + (lane >> 4), chunk slot (lane & 15) ^ (row & 15)): (1) every 8-byte cell of a pass is written once and read back as the
    element the C row expects, (2) the b128 read groups are conflict-free."""
    assert 4 * 32 * 256 == 32 * 1024
    for ih_j in itertools.product(range(2), range(8)):
        pass
    # (1) content: simulate one pass of one wave
    cell = {}
    for ih, j in itertools.product(range(2), range(8)):
        for lane in range(64):
            r16, kg = lane & 15, lane >> 4
            addr = (16 * ih + r16) * 256 + (((2 * j + (kg >> 1)) ^ r16) * 16) + 8 * (kg & 1)
            assert addr not in cell
            cell[addr] = (16 * ih + r16, 16 * j + 4 * kg)          # (row of the pass, first of 4 C columns)
    assert len(cell) == 32 * 32
    for it in range(8):
        for lane in range(64):
            row = it * 4 + (lane >> 4)
            a = row * 256 + (((lane & 15) ^ (row & 15)) * 16)
            assert cell[a] == (row, (lane & 15) * 8) and cell[a + 8] == (row, (lane & 15) * 8 + 4)
        for grp in B128_GROUPS:
            addrs = [(it * 4 + (lane >> 4)) * 256 + (((lane & 15) ^ ((it * 4 + (lane >> 4)) & 15)) * 16) for lane in grp]
            assert conflict_free(addrs, 16), (it, grp)


def test_bigd7_rings_on_512_byte_rows():
    """attn_bigd7.hip (D = 256, 64 query rows per wave, KV tiles of 32 rows x 512 B).  The kernel's address forms restated:
    K read   (kvb, ds): lane (l16, g4) -> row 16 kvb + l16, chunk 4 ds + g4 at kx[ds & 3] + (ds >> 2) * 256, kx = l16 * 512 + (((4 k4 + g4) ^ l16) * 16)
    V read   (db, x)  : lane -> kv row 16 x + 4 g4 + (l16 >> 2), 8 bytes at column 4 (l16 & 3) of pair db: vx[db & 7] + (db >> 3) * 256,
                        vx[b] = row * 512 + 8 (l16 & 3) + ((b ^ key) * 32), key = ((l16 >> 2) << 1) | (g4 & 1)
    DMA piece p = wave + 4 i (rows 2 p, 2 p + 1): lane -> row b = lane >> 5, chunk slot cs = lane & 31 <- source chunk (K) (cs & 16) | ((cs ^ (row & 15)) & 15),
                        (V) pair slot ps = cs >> 1 <- source pair (ps & 8) | ((ps ^ key(row)) & 7), key(row) = ((row & 3) << 1) | ((row >> 2) & 1)
    (1) fragment reads are bank-conflict free, (2) what the DMA writes is what the readers expect, lane offsets included."""
    ROWB = 512
    # ---- K: reads
    for ds, kvb in itertools.product(range(8), range(2)):
        for grp in B128_GROUPS:
            addrs = []
            for lane in grp:
                l16, g4 = lane & 15, lane >> 4
                a = l16 * ROWB + (((4 * (ds & 3) + g4) ^ l16) * 16) + (ds >> 2) * 256 + kvb * 16 * ROWB
                row, c = 16 * kvb + l16, 4 * ds + g4
                assert a == row * ROWB + ((c & 16) | ((c ^ (row & 15)) & 15)) * 16
                addrs.append(a)
            assert conflict_free(addrs, 16), (ds, kvb, grp)
    # ---- V: transpose reads (two 32-lane groups)
    vkey = lambda r: ((r & 3) << 1) | ((r >> 2) & 1)
    for db, x in itertools.product(range(16), range(2)):
        for grp in TR_GROUPS:
            addrs = []
            for lane in grp:
                l16, g4 = lane & 15, lane >> 4
                key = ((l16 >> 2) << 1) | (g4 & 1)
                a = (4 * g4 + (l16 >> 2)) * ROWB + 8 * (l16 & 3) + (((db & 7) ^ key) * 32) + (db >> 3) * 256 + x * 16 * ROWB
                row = 16 * x + 4 * g4 + (l16 >> 2)
                assert a == row * ROWB + ((db & 8) | ((db ^ vkey(row)) & 7)) * 32 + 8 * (l16 & 3)
                addrs.append(a)
            assert conflict_free(addrs, 8), (db, x)
    # ---- DMA: the kernel's lane offsets place source chunk / pair where the image wants them; the pieces tile the 32 rows
    rows_seen = set()
    for wave, i in itertools.product(range(4), range(4)):
        p = wave + 4 * i
        for lane in range(64):
            b, cs, ps = lane >> 5, lane & 31, (lane & 31) >> 1
            row = 2 * p + b
            rows_seen.add(row)
            # K: k_off[i & 1] = b * 512 + ((cs & 16) | ((cs ^ ((2 wave + 8 (i & 1) + b) & 15)) & 15)) * 16, source = piece base + k_off, LDS = piece base + lane * 16
            ksrc = (cs & 16) | ((cs ^ ((2 * wave + 8 * (i & 1) + b) & 15)) & 15)
            assert (ksrc & 16) | ((ksrc ^ (row & 15)) & 15) == cs                 # the image's slot of that source chunk is the lane's slot
            kv = (((2 * wave + b) & 3) << 1) | (wave >> 1)
            assert kv == vkey(row)
            vsrc_pair = (ps & 8) | ((ps ^ kv) & 7)
            assert (vsrc_pair & 8) | ((vsrc_pair ^ vkey(row)) & 7) == ps
    assert rows_seen == set(range(32))
    # ---- LDS budget: two rings of four 16-KiB tiles + query block 3's Q fragments (8 d-steps x 1 KiB x 4 waves) = 160 KiB; the epilogue's
    # staging (4 waves x 64 rows x 528 B) aliases it
    assert 2 * 4 * 32 * ROWB + 4 * 8 * 1024 == 160 * 1024 and 4 * 64 * (ROWB + 16) <= 160 * 1024


def test_bigd7_v_transposed_tile_and_permuted_k_rows():
    """attn_bigd7.hip with V as [B,H,D,N]: the LDS V tile is [256 d][32 kv] (64-B rows), 16-B chunk c of row r at slot c ^ ((-(r >> 2)) & 3);
    a Vᵀ fragment (db) is ONE ds_read_b128: lane (l16, g4) <- row 16 db + l16, chunk g4.  K: Sᵀ row l16 of kv block kvb reads K tile row
    8 (l16 >> 2) + 4 kvb + (l16 & 3) with the image key (row & 3) | ((row >> 3) & 3) << 2 — which is l16 for every such row.  Checks: both
    reads are bank-conflict free over the 4 x 16 lane groups, the DMA lane maps write what the readers expect, and the kv order a lane's P
    slots get (e = 4 kvb + r <-> kv 8 g4 + e) is the order of the Vᵀ fragment's chunk."""
    ROWB = 512
    vsw = lambda r: (-(r >> 2)) & 3
    for db in range(16):
        for grp in B128_GROUPS:
            addrs = []
            for lane in grp:
                l16, g4 = lane & 15, lane >> 4
                a = l16 * 64 + ((g4 ^ ((0 - (l16 >> 2)) & 3)) * 16) + db * 1024          # the kernel's form: vx[0] + db * 1024
                row = 16 * db + l16
                assert a == row * 64 + ((g4 ^ vsw(row)) * 16)
                addrs.append(a)
            assert conflict_free(addrs, 16), (db, grp)
    kkey = lambda r: (r & 3) | (((r >> 3) & 3) << 2)
    for ds, kvb in itertools.product(range(8), range(2)):
        for grp in B128_GROUPS:
            addrs = []
            for lane in grp:
                l16, g4 = lane & 15, lane >> 4
                a = (8 * (l16 >> 2) + (l16 & 3)) * ROWB + (((4 * (ds & 3) + g4) ^ l16) * 16) + (ds >> 2) * 256 + kvb * 4 * ROWB
                row, c = 8 * (l16 >> 2) + 4 * kvb + (l16 & 3), 4 * ds + g4
                assert kkey(row) == l16 and a == row * ROWB + ((c & 16) | ((c ^ kkey(row)) & 15)) * 16
                addrs.append(a)
            assert conflict_free(addrs, 16), (ds, kvb, grp)
    # the Sᵀ rows of a lane group: m = 4 g4 + r of block kvb stands for kv = 8 (m >> 2) + 4 kvb + (m & 3) = 8 g4 + 4 kvb + r: P slot e = 4 kvb + r
    for g4, kvb, r in itertools.product(range(4), range(2), range(4)):
        m = 4 * g4 + r
        assert 8 * (m >> 2) + 4 * kvb + (m & 3) == 8 * g4 + (4 * kvb + r)
    # DMA: K piece p = wave + 4 i (rows 2 p, 2 p + 1), key = ((2 wave + b) & 3) | i << 2; V piece p = d rows 16 p .. + 15, lane -> row lane >> 2,
    # slot lane & 3 <- source chunk slot ^ ((-(lane >> 4)) & 3)
    for wave, i in itertools.product(range(4), range(4)):
        p = wave + 4 * i
        for lane in range(64):
            b, cs = lane >> 5, lane & 31
            row = 2 * p + b
            key = ((2 * wave + b) & 3) | (i << 2)
            assert key == kkey(row)
            src = (cs & 16) | ((cs ^ key) & 15)
            assert (src & 16) | ((src ^ kkey(row)) & 15) == cs
            vrow, slot = 16 * p + (lane >> 2), lane & 3
            vsrc = slot ^ ((0 - ((lane >> 4) & 3)) & 3)
            assert vsrc ^ vsw(vrow) == slot
