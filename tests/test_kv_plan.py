"""Host logic of the frame-sink + sliding-window KV cache: the product's integer state machine
(longlive_amd/kv_cache.py) against the oracle's restatement (oracle/ref_ops.py::kv_plan, itself pinned to the
reference by the toy traces), plus domain properties at BASELINE.json's full sizes."""
import pytest
from hypothesis import given, settings, strategies as st

from longlive_amd.kv_cache import plan_update
from oracle import ref_ops as R

FS = 1560


def both(cs, n, G, E, S, sink, las, mas, recache=False):
    p = plan_update(cs, n, G, E, S, sink, las, mas, recache)
    o = R.kv_plan(cs, n, G, E, S, sink, las, mas, recache)
    assert p.current_end == o["current_end"] and p.is_recompute == o["is_recompute"]
    assert (p.local_start, p.local_end, p.write_start, p.roped_offset, p.write_len) == (
        o["local_start"], o["local_end"], o["write_start"], o["roped_offset"], o["write_len"])
    roll = None if o["roll"] is None or o["roll"]["n"] <= 0 else (o["roll"]["dst"], o["roll"]["src"], o["roll"]["n"])
    assert p.roll == roll
    assert [s for s in p.segments] == [s for s in o["segments"] if s[1] > s[0]]
    assert (p.G_new, p.E_new) == (o["G_new"], o["E_new"])
    return p


def simulate(num_blocks, nfb, window, sink, fs=FS, steps=5):
    """The pipeline's call sequence: per block `steps` forwards at the same current_start (4 denoise + clean pass).
    Tracks which absolute token sits in which slot."""
    S, sk = window * fs, sink * fs
    slots = [None] * S
    G = E = 0
    plans = []
    for b in range(num_blocks):
        cs, n = b * nfb * fs, nfb * fs
        for s in range(steps):
            p = both(cs, n, G, E, S, sk, window, S)
            if p.roll:
                d, src, cnt = p.roll
                slots[d:d + cnt] = slots[src:src + cnt]
            for i in range(p.write_len):
                slots[p.write_start + i] = cs + p.roped_offset + i
            G, E = p.G_new, p.E_new
            plans.append((b, s, p))
            # attention sees: the sink frames + the most recent (window - sink) frames incl. the current block
            seen = [slots[i] for a, e in p.segments for i in range(a, e)]
            if p.local_end < sk:
                # reference quirk (causal_model.py:334-335): while fewer tokens than the sink are cached, k[:sink]
                # still includes the not-yet-written (zero) slots.  Product == oracle was already asserted in both().
                continue
            assert None not in seen
            assert seen == sorted(seen), "keys are in temporal order"
            ce = cs + n
            expect_tail = list(range(max(sk, ce - (S - sk)) if ce > S else 0, ce))
            expect = (list(range(sk)) + [t for t in expect_tail if t >= sk]) if ce > sk else list(range(ce))
            assert seen == expect, (b, s)
    return plans, slots, (G, E)


def test_longlive_config_full_size():
    """window 12 / sink 3 / 3 frames per block, 1560 tokens per frame (configs/longlive_inference.yaml)."""
    plans, slots, (G, E) = simulate(num_blocks=8, nfb=3, window=12, sink=3)
    assert (G, E) == (8 * 3 * FS, 12 * FS)
    for b, s, p in plans:
        if b < 4:
            assert p.roll is None and p.segments == [(0, (b + 1) * 3 * FS)] or p.segments == [(0, 3 * FS), (3 * FS, (b + 1) * 3 * FS)]
        else:
            lk = sum(e - a for a, e in p.segments)
            assert lk == 18720
            if s == 0:
                assert p.roll == (4680, 9360, 9360) and not p.is_recompute          # evict 4680, keep sink
            else:
                assert p.roll is None and p.is_recompute and p.write_start == 18720 - 4680
    # sink frames are never evicted
    assert slots[:3 * FS] == list(range(3 * FS))


def test_recache_after_switch_semantics():
    """interactive_causal_inference.py:34-106: one forward over the last 12 frames, indices unchanged; with
    sink_recache the write starts at slot 0 (sink overwritten), otherwise the sink slots are protected."""
    S, sk = 12 * FS, 3 * FS
    G, E = 42 * FS, S                     # switch at frame 42 (first block start >= 40)
    p = both(30 * FS, 12 * FS, G, E, S, sk, 12, S, recache=True)
    assert p.is_recompute and p.roll is None and (p.write_start, p.roped_offset, p.write_len) == (0, 0, S)
    assert (p.G_new, p.E_new) == (G, E)
    q = both(30 * FS, 12 * FS, G, E, S, sk, 12, S, recache=False)
    assert (q.write_start, q.roped_offset, q.write_len) == (sk, sk, S - sk)
    # early switch (fewer frames than the window): current_start == 0 => not a recompute, indices re-committed
    r = both(0, 6 * FS, 6 * FS, 6 * FS, S, sk, 12, S, recache=True)
    assert not r.is_recompute and (r.G_new, r.E_new) == (6 * FS, 6 * FS)


def test_global_attention_and_no_sink():
    p = both(3 * FS, FS, 3 * FS, 3 * FS, 21 * FS, 0, -1, 32760)
    assert p.roll is None and p.segments == [(0, 4 * FS)]
    simulate(num_blocks=7, nfb=1, window=4, sink=0, fs=24, steps=3)
    simulate(num_blocks=6, nfb=2, window=5, sink=2, fs=24, steps=3)


def test_overflow_is_an_error_not_a_wild_write():
    with pytest.raises(RuntimeError, match="overflow"):
        plan_update(0, 13 * FS, 0, 0, 12 * FS, 3 * FS, -1, 32760)


@settings(max_examples=300, deadline=None)
@given(fs=st.integers(1, 40), window=st.integers(2, 9), sink=st.integers(0, 3), nfb=st.integers(1, 3),
       blocks=st.integers(1, 9), steps=st.integers(1, 3))
def test_random_geometries_match_oracle(fs, window, sink, nfb, blocks, steps):
    if sink + nfb > window:
        return
    simulate(blocks, nfb, window, sink, fs=fs, steps=steps)
