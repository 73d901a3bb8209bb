"""The segment rule of k_fk20_toom_seg (fk20_kernels.hip), restated on the host for all 686 fixed scalars of fk20_toom.h: every wavefront
derives its segment's digit positions from the scalar's digit string alone, so the K launches of a product must (a) partition the 131
positions in order, with no position done twice or skipped, and (b) carry near-equal cost -- a launch lasts as long as its slowest
piece.  (The bytes are pinned on the GPU by the ragged-batch tests; this pins the balance the timing relies on.)"""
import os
import re

HERE = os.path.dirname(os.path.abspath(__file__))
HDR = os.path.join(HERE, "..", "go-eth-kzg_amd", "csrc", "fk20_toom.h")
COST_DBL, COST_ADD, COST_TABLE = 7, 10, 90           # fk20_kernels.hip FK_SEG_COST_* (the co-Z form of round 5)


def rows():
    text = open(HDR).read()
    body = text[text.index("FK20_TOOM_WNAF[686][66]"):]
    out = []
    for m in re.finditer(r"\{((?:0x[0-9a-f]{8}u,? ?){66})\}", body):
        out.append([int(w[:-1], 16) for w in m.group(1).replace(",", " ").split()])
    assert len(out) == 686
    return out


def codes(row):
    return [(row[i >> 1] >> (16 * (i & 1))) & 0xFFFF for i in range(131)]


def segments(row, K):
    """positions (130 down to 0) -> segment, and the cost of each segment, exactly as the kernel computes them"""
    cs = codes(row)
    total, st = COST_TABLE, False
    for i in range(130, -1, -1):
        if st:
            total += COST_DBL
        for d in (cs[i] & 0xFF, cs[i] >> 8):
            if d:
                total += COST_ADD
                st = True
    seg_of, cost = {}, [0] * K
    cost[0] = COST_TABLE
    cum, st = COST_TABLE, False
    for i in range(130, -1, -1):
        pc = (COST_DBL if st else 0) + COST_ADD * (((cs[i] & 0xFF) != 0) + ((cs[i] >> 8) != 0))
        s = min(K - 1, (cum + pc // 2) * K // total)              # the segment the position's middle falls into
        seg_of[i] = s
        if st:
            cum += COST_DBL
            cost[s] += COST_DBL
        for d in (cs[i] & 0xFF, cs[i] >> 8):
            if d:
                cum += COST_ADD
                cost[s] += COST_ADD
                st = True
    assert cum == total
    return seg_of, cost, total


def test_segments_partition_the_digit_string_and_balance_the_cost():
    worst = {2: 0.0, 4: 0.0}
    for row in rows():
        for K in (2, 4):
            seg_of, cost, total = segments(row, K)
            order = [seg_of[i] for i in range(130, -1, -1)]
            assert order == sorted(order) and set(order) == set(range(K))          # in order, every segment non-empty
            assert sum(cost) == total
            worst[K] = max(worst[K], max(cost) * K / total)
    # a position costs at most 7 + 2 x 10 = 27 of ~1,450: the heaviest segment stays within a few per cent of the mean
    assert worst[2] < 1.03 and worst[4] < 1.07, worst


def test_digit_strings_are_what_the_kernel_assumes():
    """int8 digit pairs, odd digits of magnitude < 16 (the table holds 1, 3, ..., 15 times the point), nothing above position 130"""
    for row in rows():
        cs = codes(row)
        assert (row[65] >> 16) == 0
        for c in cs:
            for d in (c & 0xFF, c >> 8):
                v = d - 256 if d >= 128 else d
                assert v == 0 or (v % 2 != 0 and abs(v) <= 15)
