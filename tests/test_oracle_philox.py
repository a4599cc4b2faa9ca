"""Philox4x32-10 known-answer tests (Random123 kat_vectors) and draw conventions."""
import numpy as np

import oracle

KATS = [
    ((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
    ((0xffffffff,) * 4, (0xffffffff, 0xffffffff), (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
    ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0),
     (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1)),
]


def test_philox_kat():
    for ctr, key, exp in KATS:
        out = oracle.philox4x32_10(np.array(ctr, np.uint32), np.array(key, np.uint32))[0]
        assert tuple(int(x) for x in out) == exp


def test_u53_matches_numpy_random_sample_construction():
    # numpy legacy random_sample: (a >> 5, b >> 6) -> (a*67108864 + b) / 9007199254740992
    rs = np.random.RandomState(123)
    st = rs.get_state()
    words = np.random.RandomState(123)
    words.set_state(st)
    raw = words.randint(0, 2**32, size=8, dtype=np.uint64)  # same MT19937 words random_sample would eat
    rs2 = np.random.RandomState(123)
    rs2.set_state(st)
    u = rs2.random_sample(4)
    for k in range(4):
        assert oracle.u53(int(raw[2 * k]), int(raw[2 * k + 1])) == u[k]


def test_env_draw_counter_layout():
    seed, gid, tick, purpose = 0x1122334455667788, 0x0000000a00000005, 0x0000000300000007, 1
    w = oracle.env_draw(seed, gid, tick, purpose)
    ctr = np.array([5, 0xa, 7, 1 | (3 << 8)], np.uint32)
    key = np.array([0x55667788, 0x11223344], np.uint32)
    assert np.array_equal(w, oracle.philox4x32_10(ctr, key)[0])
