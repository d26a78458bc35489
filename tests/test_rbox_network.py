"""The 8-input compare-exchange network of rbox_device.h (fast path of box_overlap) must sort: checked exhaustively with
the 0-1 principle on the exchange list parsed from the shipped header, and — for the (key, index) order it is used
with — against a stable sort on random keys with ties."""
import itertools
import os
import re

import numpy as np

HDR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'mmdet3d-gaussian_amd', 'csrc', 'rbox_device.h')


def _network():
    text = open(HDR).read()
    body = text[text.index('#define RB_CE(i, j)'):text.index('#undef RB_CE')]
    body = body[body.index('}\n') + 1:]                     # past the macro definition
    pairs = [(int(a), int(b)) for a, b in re.findall(r'RB_CE\((\d+), (\d+)\)', body)]
    assert len(pairs) == 19 and all(0 <= a < b < 8 for a, b in pairs), pairs
    return pairs


def test_network_sorts_every_zero_one_input():
    net = _network()
    for bits in itertools.product((0, 1), repeat=8):
        v = list(bits)
        for a, b in net:
            if v[a] > v[b]:
                v[a], v[b] = v[b], v[a]
        assert v == sorted(bits), (bits, v)


def test_network_on_key_index_pairs_equals_stable_sort():
    net = _network()
    rng = np.random.default_rng(0)
    for _ in range(2000):
        cnt = int(rng.integers(1, 9))
        keys = np.round(rng.uniform(-3, 3, 8), 0 if rng.random() < 0.5 else 3).astype(np.float32)   # many ties
        if rng.random() < 0.2:
            keys[rng.integers(0, 8)] = -0.0
        keys[cnt:] = np.inf
        items = [(float(keys[k]), k) for k in range(8)]
        for a, b in net:
            ka, kb = items[a], items[b]
            if ka[0] > kb[0] or (ka[0] == kb[0] and ka[1] > kb[1]):
                items[a], items[b] = kb, ka
        want = sorted(range(cnt), key=lambda k: float(keys[k]))      # Python's sort is stable; -0.0 == 0.0
        assert [i for _, i in items[:cnt]] == want
