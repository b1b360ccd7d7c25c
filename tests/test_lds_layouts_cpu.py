"""LDS layouts of round 5 as index arithmetic: the formulas of csrc/conv.hip (convt_slab_waddr / convt_slab_read), csrc/conv_ns.hip
(NS_SW_ROW: the stride-2 halo de-interleaved by parity) and choose_conv_tile's padded x-plane stride, restated in Python and checked
for (a) being permutations that round-trip the data and (b) being free of bank conflicts under the lane groups the hardware guide
lists (ds_read_b128: four fixed 16-lane groups on 64 banks; ds_write_b64: contiguous 16-lane groups on 32 banks).  The kernels
themselves are covered by the GPU parity tests; this pins the reasoning the layouts rest on ("LDS bank conflicts")."""
import itertools

B128_GROUPS = [[0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27], [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31]]
B128_GROUPS += [[l + 32 for l in g] for g in B128_GROUPS]


def worst_multiplicity(addresses, nbytes, nbanks):
    banks = {}
    for a in addresses:
        for w in range(nbytes // 4):
            banks.setdefault((a // 4 + w) % nbanks, set()).add(a)
    return max(len(v) for v in banks.values())


# ---- transposed-conv slab (TZ = 2) ---------------------------------------------------------------------------------------------
def slab_waddr(gq, l31, t, kh):
    rec = t * 32 + (l31 ^ (4 * t))
    q = ((gq & 1) * 2 + kh) ^ ((l31 >> 2) & 3)
    return ((gq >> 1) * 64 + rec) * 32 + q * 8


def slab_read(pl, piece):
    ov, h = piece >> 1, piece & 1
    j, tz = ov >> 1, ov & 1
    sw = (j >> 2) & 3
    return (pl * 64 + tz * 32 + (j ^ (4 * tz))) * 32 + (h ^ (sw >> 1)) * 16, bool(sw & 1)


def test_convt_slab_roundtrip_and_conflicts():
    mem = {}
    for gq, l31, t, kh in itertools.product(range(4), range(32), range(2), range(2)):
        a = slab_waddr(gq, l31, t, kh)
        for b in range(8):
            assert a + b not in mem
            mem[a + b] = (gq >> 1, l31 * 2 + t, ((gq & 1) * 2 + kh) * 8 + b)   # (plane, output voxel, byte of its 32-byte record)
    assert sorted(mem) == list(range(4096))
    for pl, piece in itertools.product(range(2), range(128)):
        base, swap = slab_read(pl, piece)
        d = [mem[base + i] for i in range(16)]
        if swap:
            d = d[8:] + d[:8]
        assert d == [(pl, piece >> 1, (piece & 1) * 16 + i) for i in range(16)]
    # writes: ds_write_b64, 16 contiguous lanes (lane = kh * 32 + l31), 32 banks
    for gq, t, grp in itertools.product(range(4), range(2), range(4)):
        lanes = range(16 * grp, 16 * grp + 16)
        assert worst_multiplicity([slab_waddr(gq, l & 31, t, l >> 5) for l in lanes], 8, 32) == 1
    # reads: ds_read_b128 of pieces lane + 64 k
    for pl, k, g in itertools.product(range(2), range(2), B128_GROUPS):
        assert worst_multiplicity([slab_read(pl, l + 64 * k)[0] for l in g], 16, 64) == 1
    # the linear slab it replaces: 8-way
    old = [((0 * 64) + (l * 2 + 0)) * 32 + 0 for l in range(16)]
    assert worst_multiplicity(old, 8, 32) == 8


# ---- stride-2 halo of k_conv_ns --------------------------------------------------------------------------------------------------
def ns_slot(x, y, z):
    return (x * 9 + (y & 1) * 5 + (y >> 1)) * 24 + (z & 1) * 10 + (z >> 1)


def test_ns_stride2_halo_layout():
    slots = {ns_slot(x, y, z) for x in range(9) for y in range(9) for z in range(17)}
    assert len(slots) == 9 * 9 * 17 and max(slots) < 9 * 9 * 24          # a one-to-one map into the padded plane
    for dy, dz, g in itertools.product(range(3), range(3), B128_GROUPS[:2]):
        addrs = [ns_slot(0, 2 * (l >> 3) + dy, 2 * (l & 7) + dz) * 16 for l in g]
        assert worst_multiplicity(addrs, 16, 64) == 1
        # tap offsets are lane-independent (compile-time immediates in the kernel)
        offs = {ns_slot(0, 2 * (l >> 3) + dy, 2 * (l & 7) + dz) - ns_slot(0, 2 * (l >> 3), 2 * (l & 7)) for l in range(32)}
        assert len(offs) == 1
    linear = [((2 * (l >> 3)) * 17 + 2 * (l & 7)) * 16 for l in B128_GROUPS[0]]
    assert worst_multiplicity(linear, 16, 64) == 3                          # what the trace of round 5 paid for


# ---- k_conv_ws: x-planes padded to xs = w2 (mod 16) voxels ------------------------------------------------------------------------
def test_ws_xplane_padding_rule():
    for (w0, w2, h1, h2) in ((2, 16, 10, 18), (4, 8, 10, 10)):                # the 16^3 and 8^3 layers' M-tiles and halos
        xs = h1 * h2
        assert any(worst_multiplicity([((l // w2) * xs + l % w2) * 16 for l in g], 16, 64) > 1 for g in B128_GROUPS[:2])
        while xs % 16 != w2 % 16:
            xs += 1
        for g in B128_GROUPS[:2]:
            assert worst_multiplicity([((l // w2) * xs + l % w2) * 16 for l in g], 16, 64) == 1
