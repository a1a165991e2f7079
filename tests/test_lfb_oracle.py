"""oracle/lfb.py against a literal transcription of the reference's samplers' *structure* (CPU)."""
import numpy as np
import pytest

from oracle import lfb as ol


def _bank(rng, n_videos=3, secs=range(900, 960), dim=8, max_n=9):
    lfb = {}
    for v in range(n_videos):
        lfb[v] = {}
        for s in secs:
            n = int(rng.integers(0, max_n))
            if n:
                lfb[v][s] = [rng.standard_normal(dim) for _ in range(n)]
    return lfb


def test_construct_ava_lfb_keeps_append_order_and_rounds_keys():
    rng = np.random.default_rng(0)
    feats = [[rng.standard_normal((4, 6, 1, 1, 1))], [rng.standard_normal((3, 6, 1, 1, 1))]]
    meta = [[np.array([[7.0, 902.0, 0, 0], [7.0, 902.0, 0, 0], [8.0, 903.0, 0, 0], [7.0000001, 901.9999, 0, 0]])],
            [np.array([[8.0, 903.0, 0, 0], [7.0, 950.0, 0, 0], [7.0, 902.0, 0, 0]])]]
    lfb = ol.construct_ava_lfb(feats, meta)
    assert sorted(lfb) == [7, 8] and sorted(lfb[7]) == [902, 950]
    got = np.stack(lfb[7][902])
    want = np.stack([feats[0][0][0], feats[0][0][1], feats[0][0][3], feats[1][0][2]]).reshape(4, 6)
    assert np.array_equal(got, want)
    assert len(lfb[8][903]) == 2


def test_sample_lfb_ava_structure_matches_reference_semantics():
    rng = np.random.default_rng(1)
    lfb = _bank(rng)
    W, K, D = 20, 5, 8
    for video in lfb:
        for sec in (905, 930, 959, 1200):
            out = ol.sample_lfb_ava(lfb[video], sec, W, K, D, seed=11, sample_id=3, video=video)
            assert out.shape == (W * K, D)
            lower = sec - W // 2
            for j in range(W):
                si = lower + j
                rows = out[j * K:(j + 1) * K]
                n = len(lfb[video].get(si, []))
                m = min(n, K)
                assert np.all(rows[m:] == 0)
                pool = [tuple(f) for f in lfb[video].get(si, [])]
                picked = [tuple(r) for r in rows[:m]]
                assert len(set(picked)) == m and all(p in pool for p in picked)     # distinct, from this second
    # same draw id -> same sample; another id -> (almost surely) another
    a = ol.sample_lfb_ava(lfb[0], 930, W, K, D, 11, 3, 0)
    assert np.array_equal(a, ol.sample_lfb_ava(lfb[0], 930, W, K, D, 11, 3, 0))
    assert not np.array_equal(a, ol.sample_lfb_ava(lfb[0], 930, W, K, D, 11, 4, 0))


def test_choice_is_close_to_uniform():
    n, k, trials = 7, 3, 4000
    first = np.zeros(n)
    member = np.zeros(n)
    for t in range(trials):
        c = ol.choice_without_replacement(n, k, seed=5, sample_id=t, video=1, step=40)
        assert len(set(c)) == k
        first[c[0]] += 1
        member[c] += 1
    assert np.all(np.abs(first / trials - 1.0 / n) < 0.03)
    assert np.all(np.abs(member / trials - float(k) / n) < 0.04)


def test_charades_window_and_compaction():
    rng = np.random.default_rng(2)
    frames = ol.charades_lfb_frames([100, 30, 400], clips_per_second=2)
    assert frames[0] == (0, 11) and all((f + 1) % 12 == 0 for _, f in frames) and (1, 23) in frames
    feats = rng.standard_normal((len(frames) + 3, 4, 1, 1, 1))          # 3 padding rows of the last batch
    lfb = ol.construct_frame_level_lfb([[feats[:10]], [feats[10:]]], frames)
    assert sorted(lfb) == [0, 1, 2] and len(lfb[2]) == 33
    assert ol.charades_window(200, 20, 2) == (80, 320)
    out = ol.sample_lfb_charades(lfb[2], 200, 20, 2, 4)
    want = [lfb[2][f] for f in sorted(lfb[2]) if 80 <= f <= 320][:20]
    assert len(want) == 20 and np.array_equal(out, np.array(want))
    # clip near the start: fewer than `window` frames, zero tail
    out = ol.sample_lfb_charades(lfb[1], 10, 20, 2, 4)
    have = [f for f in sorted(lfb[1]) if -110 <= f <= 130]
    assert np.array_equal(out[:len(have)], np.array([lfb[1][f] for f in have])) and np.all(out[len(have):] == 0)


def test_device_bank_step_window_equals_the_reference_frame_window():
    """host index math of DeviceBank.sample_frames (no GPU): bank steps [lo, hi] are exactly the LFB
    frames the reference would find in [begin, end]"""
    from vlfb.lfb_bank import frame_window_steps
    for cps in (1, 2, 3, 4):
        freq = ol.FPS // cps
        for window in (4, 10, 20):
            centers = np.arange(-50, 900, 7)
            lo, hi = frame_window_steps(centers, window, cps)
            for c, l, h in zip(centers, lo, hi):
                begin, end = ol.charades_window(int(c), window, cps)
                want = [f for f in range(begin, end + 1) if f >= 0 and (f + 1) % freq == 0]
                got = [freq * (t + 1) - 1 for t in range(max(int(l), 0), int(h) + 1)]
                assert got == want, (cps, window, c)


def test_epic_window_arithmetic_matches_the_reference_loops():
    """host index math of DeviceBank.sample_epic_verb / sample_epic_noun (bank steps that lie inside the frame
    window) against the frame loops of epic.py:310-374 restated in oracle/lfb.py, incl. negative windows (Python
    int() truncates toward zero) and clip centres at the start of a video"""
    from vlfb.lfb_bank import epic_verb_window_steps, epic_noun_window_steps
    rng = np.random.default_rng(5)
    for window in (40, 30, 7):
        centres = np.concatenate([rng.integers(0, 20000, 40), np.arange(0, 40), [299, 300, 301, 599, 600]])
        lo, hi = epic_verb_window_steps(centres, window)
        for c, l, h in zip(centres, lo, hi):
            half = (window * ol.EPIC_FPS) // 2
            frames = [f for f in range(int(c) - half, int(c) + half + 1) if f % 30 == 0]
            want = (frames[0] // 30, frames[-1] // 30) if frames else None
            if want is None:
                assert l > h
            else:
                assert (l, h) == want, (window, c)
        lo, hi = epic_noun_window_steps(centres, window)
        for c, l, h in zip(centres, lo, hi):
            secs = float(window) / 10
            lower = int(c - (secs / 2) * 30)
            upper = int(lower + secs * 30)
            frames = [f for f in range(lower, upper + 1) if f % 30 == 0]
            if frames:
                assert (l, h) == (frames[0] // 30, frames[-1] // 30), (window, c)
            else:
                assert l > h


def test_reference_draw_table_replays_sample_lfb_including_a_repeated_keyframe():
    """lfb_bank.reference_draw_table (host half of DeviceBank.sample_window_reference_draw) against the reference's own loop
    (lib/datasets/ava.py:300-323 `sample_lfb`, called once per clip of the minibatch, ava.py:230): same np.random stream, same
    slots -- also when one keyframe sits in the minibatch TWICE (two draws, not one) and when a clip has several boxes (one
    draw, repeated per box: ava_data_input.py:191-192)."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "video-long-term-feature-banks_amd", "lib"))
    from vlfb.lfb_bank import reference_draw_table
    W, K = 5, 3
    gen = np.random.RandomState(7)
    counts = gen.randint(0, 6, size=(4, 40)).astype(np.int64)          # features stored per (video, second)
    counts[2, 10:13] = 0                                               # seconds without features: `si not in in_video_lfb`
    clips = [(1, 20), (2, 11), (1, 20), (3, 1), (0, 38)]               # (video, sec); clip 0 and clip 2: the SAME keyframe
    boxes = [2, 1, 3, 1, 2]                                            # rows per clip

    def sample_lfb_slots(video, sec, rng):                             # the reference loop, slots instead of feature copies
        t = np.full((W, K), -1, dtype=np.int32)
        lower = sec - W // 2
        for j, si in enumerate(range(lower, lower + W)):
            n = int(counts[video, si]) if 0 <= si < counts.shape[1] else 0
            if n > 0:
                used = min(n, K)
                t[j, :used] = rng.choice(range(n), used, replace=False)
        return t
    ref_rng = np.random.RandomState(123)
    want = []
    for (v, sec), nb in zip(clips, boxes):
        t = sample_lfb_slots(v, sec, ref_rng)                          # one draw per clip of the minibatch ...
        want += [t] * nb                                               # ... repeated for every box of the clip
    vid = np.concatenate([[v] * nb for (v, _), nb in zip(clips, boxes)])
    centre = np.concatenate([[s_] * nb for (_, s_), nb in zip(clips, boxes)])
    clip_index = np.concatenate([[i] * nb for i, nb in enumerate(boxes)])
    got = reference_draw_table(counts, vid, centre, clip_index, W, K, np.random.RandomState(123))
    assert np.array_equal(got, np.stack(want))
    assert not np.array_equal(got[0], got[3])                          # the repeated keyframe was drawn again
    # keyframe ids instead of minibatch positions (the two equal keyframes would share one draw): refused
    with pytest.raises(ValueError):
        reference_draw_table(counts, vid, centre, np.concatenate([[7] * 2, [9], [7] * 3, [4], [5] * 2]), W, K, np.random.RandomState(123))
