"""Philox-4x32-10 known answers (Random123 kat_vectors: three published vectors)."""
import numpy as np

from oracle import oracle as O

KAT = [
    ([0, 0, 0, 0], [0, 0], [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]),
    ([0xffffffff] * 4, [0xffffffff] * 2, [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]),
    ([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], [0xa4093822, 0x299f31d0],
     [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]),
]


def test_philox_kat():
    for ctr, key, want in KAT:
        assert [int(x) for x in O.philox4x32(ctr, key)] == want


def test_philox_counter_sensitivity():
    a = O.philox4x32([1, 2, 3, 4], [5, 6])
    for i in range(4):
        c = [1, 2, 3, 4]
        c[i] += 1
        assert not np.array_equal(a, O.philox4x32(c, [5, 6]))
