"""The feature-bank KERNELS against outputs of the reference's own bank code.

tests/golden/ref_aux.npz holds what tools/lfb_loader.py (construct_ava_lfb, construct_frame_level_lfb) and the samplers of
lib/datasets/{ava,charades,epic}.py returned on synthetic banks when oracle/make_ref_aux_golden.py ran them from
/root/reference.  Here the same feature batches go through vlfb.lfb_bank.DeviceBank (vlfb_lfb_append,
vlfb_lfb_sample_compact / _packed / _window in csrc/vlfb_lfb.hip; fp32 bank, fp32 output) and must give the same banks and
the same samples, exactly.  The AVA draw exists twice: a counter-based key on the device (sample_window: the reference's
distribution, another stream -- occupied rows, their second and their distinctness are compared) and the reference's own
np.random.choice stream drawn on the host with the device only gathering (sample_window_reference_draw: exact)."""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

Z = np.load(os.path.join(os.path.dirname(__file__), "golden", "ref_aux.npz"))
META = json.loads(bytes(Z["meta"]).decode())


def _case(kind):
    return [c for c in META["lfb"] if c["kind"] == kind][0]


def test_charades_bank_built_and_sampled_on_the_device():
    from vlfb import lfb_bank as lb
    case = _case("charades")
    D, per, cps = case["dim"], case["per_gpu"], case["clips_per_second"]
    sf = lb.FPS // cps
    frames = [tuple(int(x) for x in f) for f in Z["lfb_ch_frames"]]
    n_steps = max(case["num_frames"]) // sf
    bank = lb.DeviceBank(len(case["num_frames"]), n_steps, 1, D, "fp32", "cuda:0")
    flat = Z["lfb_ch_feats"]
    g = 0
    for b in range(len(flat) // per):                        # one per-GPU batch of one inference iteration at a time
        batch = torch.as_tensor(flat[b * per:(b + 1) * per].reshape(per, D, 1, 1, 1))
        bank.append_frames(batch, frames[g:g + per], sf)     # (rows past the last LFB frame are the padding of the last batch)
        g += per
    bank.check_no_drops()
    ref = bank.to_reference(frame_level=True, sample_freq=sf)
    keys = [(v, f) for v in sorted(ref) for f in sorted(ref[v])]
    assert np.array_equal(np.array(keys), Z["lfb_ch_bank_keys"])
    assert np.array_equal(np.array([ref[v][f] for v, f in keys], dtype=np.float32), Z["lfb_ch_bank_rows"])
    q = Z["lfb_ch_queries"]
    out = bank.sample_frames(q[:, 0], q[:, 1], case["window"], cps, out_dtype=torch.float32).cpu().numpy()
    assert np.array_equal(out.astype(np.float64), Z["lfb_ch_samples"])


def test_ava_bank_built_on_the_device_and_the_draw_has_the_reference_structure():
    from vlfb import lfb_bank as lb
    case = _case("ava")
    D, K, W = case["dim"], case["max_per_step"], case["window"]
    meta = Z["lfb_ava_meta"]
    secs = np.round(meta[:, 1]).astype(np.int64)
    vids = np.round(meta[:, 0]).astype(np.int64)
    cap = int(max(np.unique(np.stack([vids, secs], 1), axis=0, return_counts=True)[1]))
    bank = lb.DeviceBank(int(vids.max()) + 1, int(secs.max() - secs.min() + 1), cap, D, "fp32", "cuda:0", int(secs.min()))
    at = 0
    for n in Z["lfb_ava_batch_rows"].reshape(-1):            # batches in the order the loop of lfb_loader.py:84-86 visits them
        n = int(n)
        bank.append_ava(torch.as_tensor(Z["lfb_ava_feats"][at:at + n].reshape(n, D, 1, 1, 1)), meta[at:at + n])
        at += n
    bank.check_no_drops()
    ref = bank.to_reference()
    keys, rows = [], []
    for v in sorted(ref):
        for s in sorted(ref[v]):
            for f in ref[v][s]:
                keys.append((v, s))
                rows.append(f)
    assert np.array_equal(np.array(keys), Z["lfb_ava_bank_keys"])             # append order inside a second kept
    assert np.array_equal(np.array(rows, dtype=np.float32), Z["lfb_ava_bank_rows"])
    for i, d in enumerate(case["draws"]):
        out = bank.sample_window([d["video"]], [d["sec"]], [i], W, K, 1234, out_dtype=torch.float32).cpu().numpy()[0]
        want = Z["lfb_ava_sample_%d" % i]
        assert out.shape == want.shape
        assert np.array_equal(np.any(out != 0, 1), np.any(want != 0, 1)), i   # same rows occupied, same zero padding
        lower = d["sec"] - W // 2
        for j in range(W):
            have = [tuple(r) for r in out[j * K:(j + 1) * K] if np.any(r != 0)]
            pool = [tuple(np.float32(f)) for f in ref.get(d["video"], {}).get(lower + j, [])]
            assert len(set(have)) == len(have) and all(h in pool for h in have), (i, j)


def test_ava_window_with_the_references_own_random_stream_is_the_references_sample():
    """DeviceBank.sample_window_reference_draw (vlfb_lfb_gather_slots): the host makes the np.random.choice calls of
    ava.py:316-318 in the reference's order, the device gathers -- seeded as the fixture generator seeded the reference
    (np.random.seed(np_seed) in front of ava.sample_lfb), the sampled bank IS the reference's, element for element; two boxes
    of one clip share the draw (ava_data_input.py:191-192)"""
    from vlfb import lfb_bank as lb
    case = _case("ava")
    D, K, W = case["dim"], case["max_per_step"], case["window"]
    meta = Z["lfb_ava_meta"]
    secs = np.round(meta[:, 1]).astype(np.int64)
    vids = np.round(meta[:, 0]).astype(np.int64)
    cap = int(max(np.unique(np.stack([vids, secs], 1), axis=0, return_counts=True)[1]))
    bank = lb.DeviceBank(int(vids.max()) + 1, int(secs.max() - secs.min() + 1), cap, D, "fp32", "cuda:0", int(secs.min()))
    at = 0
    for n in Z["lfb_ava_batch_rows"].reshape(-1):
        n = int(n)
        bank.append_ava(torch.as_tensor(Z["lfb_ava_feats"][at:at + n].reshape(n, D, 1, 1, 1)), meta[at:at + n])
        at += n
    for i, d in enumerate(case["draws"]):
        np.random.seed(d["np_seed"])
        out = bank.sample_window_reference_draw([d["video"], d["video"]], [d["sec"], d["sec"]], [0, 0], W, K,
                                                out_dtype=torch.float32).cpu().numpy()
        want = Z["lfb_ava_sample_%d" % i]
        assert np.array_equal(out[0].astype(np.float64), want), i
        assert np.array_equal(out[1], out[0])              # the second box of the clip: the same bank, no second draw
    # a table entry outside the occupied slots reads as zeros, not as stale memory
    rng = np.random.RandomState(0)
    out = bank.sample_window_reference_draw([0], [int(secs.min()) - 50], [0], W, K, rng=rng, out_dtype=torch.float32)
    assert float(out.abs().sum()) == 0.0


def test_epic_banks_sampled_on_the_device():
    from vlfb import lfb_bank as lb
    case = _case("epic_verb")
    verb = {0: {int(f): r for f, r in zip(Z["lfb_eva_bank_keys"], Z["lfb_eva_bank_rows"])}}
    bank = lb.DeviceBank.from_epic(verb, noun=False, dtype="fp32")
    q = Z["lfb_eva_queries"]
    out = bank.sample_epic_verb([0] * len(q), q, case["window"], out_dtype=torch.float32).cpu().numpy()
    assert np.array_equal(out, Z["lfb_eva_samples"].astype(np.float32))
    case = _case("epic_noun")
    noun, at = {0: {}}, 0
    for f, n in Z["lfb_en_counts"]:
        noun[0][int(f)] = Z["lfb_en_rows"][at:at + n] if n else []
        at += int(n)
    bank = lb.DeviceBank.from_epic(noun, noun=True, sample_freq=lb.EPIC_FPS // case["frames_per_second"], dtype="fp32")
    q = Z["lfb_en_queries"]
    out = bank.sample_epic_noun([0] * len(q), q, case["window"], case["max_per_frame"], case["frames_per_second"],
                                out_dtype=torch.float32).cpu().numpy()
    assert np.array_equal(out.astype(np.float64), Z["lfb_en_samples"])
