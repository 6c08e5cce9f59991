"""The offset groups of the weight-gradient kernel for 3 x 3 x 3 products (csrc/spconv_dw2.hip, spconv_dw3_kernel): the
four packed tables must partition the 27 offsets, hold at most 8 each (a wave has 8 accumulators), and no group may take
more than 3 of the 9 offsets of an axis-aligned plane (centre + 4 faces + 4 edges: the activity pattern of a flat
surface) -- the property the partition was chosen for (tools/dw_balance.py measures it on the bench scene)."""
import os
import re

SRC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "3d-wsis_amd", "csrc", "spconv_dw2.hip")


def _groups():
    text = open(SRC).read()
    block = text[text.index("const unsigned long long ktab ="):]
    block = block[:block.index(";") + 1]
    consts = re.findall(r"0x([0-9a-f]{16})ull", block)
    assert len(consts) == 4, consts
    groups = []
    for c in consts:
        v = int(c, 16)
        slots = [(v >> (8 * j)) & 0xff for j in range(8)]
        live = [k for k in slots if k != 0xff]
        assert slots[:len(live)] == live, "unused slots come last"
        assert live == sorted(live), "slots in ascending offset order"
        groups.append(live)
    return groups


def test_offset_groups_partition_the_kernel():
    g = _groups()
    assert sorted(k for grp in g for k in grp) == list(range(27))
    assert all(1 <= len(grp) <= 8 for grp in g)


def test_no_group_takes_more_than_three_offsets_of_an_axis_plane():
    g = _groups()
    offs = [(a, b, c) for a in range(3) for b in range(3) for c in range(3)]      # k = 9a + 3b + c
    for axis in range(3):
        plane = {k for k, o in enumerate(offs) if o[axis] == 1}
        assert len(plane) == 9
        loads = [len(plane & set(grp)) for grp in g]
        assert sum(loads) == 9 and max(loads) <= 3, (axis, loads)
    centre = [i for i, grp in enumerate(g) if 13 in grp]
    assert len(centre) == 1
    # the centre's group holds the fewest face / edge offsets: it is active in every slice
    cls = lambda k: sum(1 for v in offs[k] if v != 1)
    heavy = [sum(1 for k in grp if cls(k) in (1, 2)) for grp in g]
    assert heavy[centre[0]] == min(heavy)
