"""CPU model of the 5-NN walk's truncated-key top list (msf_loam_amd/csrc/msfl_kernels.cuh: Top6K, top_insert_off, top_settled).

The HIP kernels keep six sorted 32-bit keys (f32 distance bits & ~7) | slot, store each kept candidate's position in a slot
table and search a query again with exact 64-bit (distance, index) keys when `top_settled` is false.  This test restates
that list in numpy integer arithmetic and checks, on adversarial streams (exact ties, distances one ulp apart, values at
the gate, fewer than five candidates, sorted / reversed / random arrival), the property the kernels rely on:

    whenever the model says "settled", the five positions it reports, in its order, are exactly the five smallest
    candidates by (distance, index) among those with distance <= gate, and "found" is exactly "five such candidates
    exist and the fifth is strictly below the gate".

Nothing here touches the GPU or the oracle library; the GPU-side counterpart is
tests/test_gpu_scan2map.py::test_truncated_key_walk_is_exact_on_ties_and_at_the_gate."""
import numpy as np
import pytest

SENT = 0xFFFFFF00


def f32_bits(x):
    return np.asarray(x, dtype=np.float32).view(np.uint32)


def med3(a, b, c):
    return sorted((a, b, c))[1]


class Top6K:
    """Bit-for-bit restatement of the device structure (unsigned 32-bit arithmetic in Python ints)."""

    def __init__(self, gate_bits):
        self.k = [SENT + 9 * i for i in range(6)]
        self.gate = int(gate_bits)
        self.bound = self.gate
        self.slots = [None] * 8

    def insert(self, d_bits, pos):
        d_bits = int(d_bits)
        if d_bits > self.bound:                      # pre-filter
            return
        k = self.k
        slot = k[5] & 7
        x = (d_bits & ~7 & 0xFFFFFFFF) | slot
        self.slots[slot] = pos                       # ds_write_b32 before the chain: may overwrite the 6th key's position
        k5 = med3(k[4], k[5], x); k4 = med3(k[3], k[4], x); k3 = med3(k[2], k[3], x)
        k2 = med3(k[1], k[2], x); k1 = med3(k[0], k[1], x); k0 = min(k[0], x)
        self.k = [k0, k1, k2, k3, k4, k5]
        self.bound = min(k4 | 7, self.gate)

    def settled(self, accept_gate_bits):
        k = self.k
        amb = min(k[0] ^ k[1], k[1] ^ k[2], k[2] ^ k[3], k[3] ^ k[4], k[4] ^ k[5], k[4] ^ int(accept_gate_bits))
        return amb >= 8

    def found(self):
        return self.k[4] < SENT

    def positions(self):
        return [self.slots[k & 7] for k in self.k[:5]]


def exact_top5(d_bits, gate_bits):
    """(found, positions) by the reference rule: candidates with d <= gate ordered by (d, index); accepted if the 5th < gate."""
    idx = [i for i in range(len(d_bits)) if int(d_bits[i]) <= gate_bits]
    idx.sort(key=lambda i: (int(d_bits[i]), i))
    if len(idx) < 5 or not int(d_bits[idx[4]]) < gate_bits:
        return False, None
    return True, idx[:5]


def streams(rng):
    gate = np.float32(1.0)
    for case in range(400):
        n = int(rng.integers(0, 40))
        kind = case % 8
        if kind == 0:                                   # plain random distances, some beyond the gate
            d = rng.uniform(0.0, 1.6, n).astype(np.float32)
        elif kind == 1:                                 # few distinct values: exact ties everywhere
            d = rng.choice(np.array([0.0625, 0.25, 0.3125, 0.5, 1.0, 1.5], np.float32), n)
        elif kind == 2:                                 # distances a few ulps apart (inside one 8-ulp bucket and across its edges)
            base = f32_bits(np.float32(rng.uniform(0.05, 0.9)))
            d = (int(base) + rng.integers(-12, 13, n)).astype(np.uint32).view(np.float32)
        elif kind == 3:                                 # around the gate: below, at, above, within its bucket
            d = (int(f32_bits(gate)) + rng.integers(-10, 11, n)).astype(np.uint32).view(np.float32)
            d = np.concatenate([d, rng.uniform(0, 0.5, int(rng.integers(0, 6))).astype(np.float32)])
        elif kind == 4:                                 # ascending arrival (every candidate inserts), then shuffled below
            d = np.sort(rng.uniform(0.0, 1.2, n).astype(np.float32))[::-1].copy()
        elif kind == 5:                                 # well separated: must settle
            d = (np.arange(n, dtype=np.float32) * np.float32(0.03) + np.float32(0.01))[rng.permutation(n)]
        elif kind == 6:                                 # zeros and denormals
            d = rng.choice(np.array([0.0, 1e-45, 2e-45, 1e-38, 0.5], np.float32), n)
        else:                                           # a tie exactly at the 5th / 6th boundary inside otherwise separated values
            d = (np.arange(n, dtype=np.float32) * np.float32(0.04) + np.float32(0.02))
            if n >= 7:
                d[5] = d[4]
            d = d[rng.permutation(n)]
        yield d.astype(np.float32), gate


def test_settled_lists_are_the_exact_top_five():
    rng = np.random.default_rng(20260930)
    n_settled = n_redo = n_found = 0
    for d, gate in streams(rng):
        bits = f32_bits(d)
        gate_bits = int(f32_bits(gate))
        t = Top6K(gate_bits)
        for pos, b in enumerate(bits):
            t.insert(b, pos)
        ok, want = exact_top5(bits, gate_bits)
        if t.settled(gate_bits):
            n_settled += 1
            assert t.found() == ok, (d, t.k)
            if ok:
                n_found += 1
                assert t.positions() == want, (d, t.k, t.positions(), want)
        else:
            n_redo += 1                                  # the kernels search these again with the exact keys: nothing to check but
            # that the redo was called for by a real ambiguity: two kept distances (or the 5th and the gate) in one 8-ulp bucket
            k = t.k
            close = min(k[0] ^ k[1], k[1] ^ k[2], k[2] ^ k[3], k[3] ^ k[4], k[4] ^ k[5], k[4] ^ gate_bits)
            assert close < 8
    assert n_settled > 150 and n_found > 80 and n_redo > 50, (n_settled, n_found, n_redo)


def test_separated_distances_always_settle():
    """Distances that differ by more than one bucket (8 ulps) never need the exact redo, whatever the arrival order."""
    rng = np.random.default_rng(7)
    for _ in range(100):
        n = int(rng.integers(5, 30))
        bits = (int(f32_bits(np.float32(0.01))) + 16 * rng.permutation(200)[:n] * 50).astype(np.uint32)
        gate_bits = int(f32_bits(np.float32(1.0)))
        t = Top6K(gate_bits)
        for pos, b in enumerate(bits):
            t.insert(b, pos)
        assert t.settled(gate_bits)
        ok, want = exact_top5(bits, gate_bits)
        assert t.found() == ok and (not ok or t.positions() == want)


@pytest.mark.parametrize("seed_bits", [0x3F000000, 0x3E99999A])
def test_seeded_initial_bound_keeps_the_result(seed_bits):
    """The second outer iteration may start from a bound tighter than the gate that five real candidates are known to meet
    (knn5_seed_bound): the list then never sees candidates beyond it, and the result is still the exact top five."""
    rng = np.random.default_rng(seed_bits)
    gate_bits = int(f32_bits(np.float32(1.0)))
    for _ in range(100):
        n = int(rng.integers(8, 40))
        d = rng.uniform(0.0, 1.2, n).astype(np.float32)
        bits = f32_bits(d)
        if np.sum(bits <= seed_bits) < 5:
            continue
        t = Top6K(seed_bits)
        for pos, b in enumerate(bits):
            t.insert(b, pos)
        if t.settled(gate_bits):
            ok, want = exact_top5(bits, gate_bits)
            assert ok and t.found() and t.positions() == want
