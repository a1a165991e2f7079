"""ORACLE (test infrastructure only -- never imported by the product path).

CPU restatement of the reference's long-term-feature-bank construction and sampling:
  * construct_ava_lfb            tools/lfb_loader.py:79-112   bank[video][sec] = [feat, ...] in append order
  * construct_frame_level_lfb    tools/lfb_loader.py:49-76    bank[video][frame] = feat
  * sample_lfb (AVA)             lib/datasets/ava.py:300-323
  * sample_lfb (Charades)        lib/datasets/charades.py:251-276
  * sample_verb_lfb / sample_noun_lfb (EPIC-Kitchens)   lib/datasets/epic.py:310-374

Pinned to the reference's own functions, executed on synthetic banks (oracle/make_ref_aux_golden.py ->
tests/golden/ref_aux.npz, tests/test_ref_aux.py): construction, the Charades / EPIC windows and -- with
the draw below swapped for the reference's `np.random.choice` under the same seed -- the AVA sampler, all
bit for bit.  Not replayed: the draw itself.  The reference takes `np.random.choice(..., replace=False)`
from the global MT19937 stream, which no device kernel can (or should) follow.  What is restated exactly
is the *structure* -- which rows of the output
belong to which time step, how many features a step contributes, distinctness, zero padding, the
Charades window arithmetic -- and the draw itself is replaced by the counter-based key
`choice_key` shared bit for bit with csrc/vlfb_lfb.hip: the chosen features of a step are the
min(n, K) smallest keys, in key order (a uniform random ordered subset, like the reference's).
"""
import numpy as np

FPS = 24     # lib/datasets/charades.py:43


def _mix32(x):
    x &= 0xFFFFFFFF
    x ^= x >> 16
    x = (x * 0x85EBCA6B) & 0xFFFFFFFF
    x ^= x >> 13
    x = (x * 0xC2B2AE35) & 0xFFFFFFFF
    x ^= x >> 16
    return x


def choice_key(seed, sample_id, video, step, i):
    h = _mix32((seed & 0xFFFFFFFF) ^ (sample_id & 0xFFFFFFFF))
    h = _mix32((h + 0x9E3779B9 + (((seed >> 32) & 0xFFFFFFFF) ^ (video & 0xFFFFFFFF))) & 0xFFFFFFFF)
    h = _mix32((h + 0x85EBCA6B + (step & 0xFFFFFFFF)) & 0xFFFFFFFF)
    h = _mix32((h + 0xC2B2AE35 + (i & 0xFFFFFFFF)) & 0xFFFFFFFF)
    return h


def choice_without_replacement(n, k, seed, sample_id, video, step):
    """ordered k-subset of range(n): stands in for np.random.choice(range(n), k, replace=False)"""
    keys = sorted((choice_key(seed, sample_id, video, step, i), i) for i in range(n))
    return [i for _, i in keys[:k]]


def construct_ava_lfb(all_features, all_metadata):
    """lfb_loader.py:79-112.  all_features: per iteration a list (per GPU) of (R, D[,1,1,1]) arrays;
    all_metadata: same nesting of (R, 4) arrays [video_id, sec, ...]."""
    lfb = {}
    for iter_features, iter_metadata in zip(all_features, all_metadata):
        for gpu_features, gpu_metadata in zip(iter_features, iter_metadata):
            assert gpu_features.shape[0] == gpu_metadata.shape[0]
            for i in range(gpu_features.shape[0]):
                video_id = int(np.round(gpu_metadata[i][0]))
                sec = int(np.round(gpu_metadata[i][1]))
                lfb.setdefault(video_id, {}).setdefault(sec, []).append(np.squeeze(gpu_features[i]))
    return lfb


def construct_frame_level_lfb(all_features, all_metadata):
    """lfb_loader.py:49-76 (Charades flavour: metadata rows are (video_id, frame_id)); features past
    the end of the metadata list are the padding of the last partial batch and are dropped."""
    lfb = {}
    g = 0
    for iter_features in all_features:
        for gpu_features in iter_features:
            for i in range(gpu_features.shape[0]):
                if g >= len(all_metadata):
                    break
                video_id, frame_id = all_metadata[g]
                g += 1
                lfb.setdefault(video_id, {})[frame_id] = np.squeeze(gpu_features[i])
    return lfb


def sample_lfb_ava(in_video_lfb, sec, window_size, max_per_step, dim, seed, sample_id, video):
    """ava.py:300-323 -> (window_size * max_per_step, dim)"""
    K = max_per_step
    lower = sec - (window_size // 2)
    out = np.zeros((window_size * K, dim), dtype=np.float64)
    for j, si in enumerate(range(lower, lower + window_size)):
        if si in in_video_lfb:
            n = len(in_video_lfb[si])
            for k, idx in enumerate(choice_without_replacement(n, min(n, K), seed, sample_id, video, si)):
                out[j * K + k] = in_video_lfb[si][idx]
    return out


def charades_window(center_idx, window_size, clips_per_second):
    """first and last frame (inclusive) searched around `center_idx` (charades.py:259-261)"""
    secs = window_size // clips_per_second
    begin = int(np.round(center_idx - (float(secs) / 2.0 * FPS)))
    return begin, begin + secs * FPS


def sample_lfb_charades(video_lfb, center_idx, window_size, clips_per_second, dim):
    """charades.py:251-276 -> (window_size, dim); an empty window yields zeros (the reference logs)"""
    begin, end = charades_window(center_idx, window_size, clips_per_second)
    rows = []
    for frame_idx in range(begin, end + 1):
        if frame_idx in video_lfb and len(rows) < window_size:
            rows.append(video_lfb[frame_idx])
    out = np.zeros((window_size, dim), dtype=np.float64)
    if rows:
        out[:len(rows)] = np.array(rows)
    return out


def charades_lfb_frames(num_frames_per_video, clips_per_second):
    """(video_idx, frame) pairs the bank is inferred on (charades.py:238-248)"""
    sample_freq = FPS // clips_per_second
    return [(v, i) for v, n in enumerate(num_frames_per_video) for i in range(n) if (i + 1) % sample_freq == 0]


EPIC_FPS = 30    # lib/datasets/epic.py:46


def sample_verb_lfb_epic(center_idx, video_lfb, window_size, dim):
    """epic.py:310-331 -> (window_size, dim)"""
    half_len = (window_size * EPIC_FPS) // 2
    rows = []
    for frame_idx in range(center_idx - half_len, center_idx + half_len + 1):
        if frame_idx in video_lfb and len(rows) < window_size:
            rows.append(video_lfb[frame_idx])
    out = np.zeros((window_size, dim), dtype=np.float64)
    if rows:
        out[:len(rows)] = np.array(rows)
    return out


def sample_noun_lfb_epic(center_idx, video_lfb, window_size, dim, max_per_frame=10, frames_per_second=1):
    """epic.py:338-374 -> (window_size, dim); `video_lfb[frame]` is an (n, dim) array or []"""
    secs = float(window_size) / (max_per_frame * frames_per_second)
    lower = int(center_idx - (secs / 2) * EPIC_FPS)
    upper = int(lower + secs * EPIC_FPS)
    chunks, num_feat = [], 0
    for frame_idx in range(lower, upper + 1):
        if frame_idx in video_lfb:
            frame_lfb = video_lfb[frame_idx]
            if not (isinstance(frame_lfb, list) and len(frame_lfb) == 0):
                curr = min(max_per_frame, frame_lfb.shape[0])
                num_feat += curr
                chunks.append(frame_lfb[:curr])
                if num_feat >= window_size:
                    break
    out = np.zeros((window_size, dim), dtype=np.float64)
    if chunks:
        got = np.vstack(chunks)[:window_size]
        out[:got.shape[0]] = got
    return out
