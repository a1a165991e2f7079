"""The long-term feature bank as a device tensor.

Reference: tools/lfb_loader.py builds `{video: {sec: [feat, ...]}}` (AVA, :79-112) or
`{video: {frame: feat}}` (Charades/EPIC, :49-76) on the host from `box_pooled` / `pool5` fetched
after every inference iteration, pickles it, and the dataset classes sample a window per clip in
NumPy (lib/datasets/ava.py:300-323, charades.py:251-276) -- 2.46 MB per RoI through the blob queue
every training iteration.

Here the bank lives in HBM for the whole job (AVA train: 235 videos x 900 s x 16 slots x 2048 bf16
= 13.9 GB of the 288 GB), `box_pooled` is appended to it by a kernel without leaving the device,
and the sampled (R, K, 2048) block is written by a kernel straight into the model's `lfb` input.
`from_reference` / `to_reference` convert to and from the reference's pickled dictionaries.

All compute goes through libvlfb_hip.so (vlfb_lfb_*); there is no host fallback.
"""
import ctypes as C

import numpy as np
import torch

from vlfb import hip

FPS = 24   # cfg.CHARADES.FPS (lib/datasets/charades.py:43)


def frame_window_steps(center_frames, window, clips_per_second):
    """first / last bank step (inclusive) of the frame window the reference searches around each clip
    centre (charades.py:259-261: begin = round(centre - secs/2 * FPS), end = begin + secs * FPS).
    Bank step t holds frame sample_freq * (t + 1) - 1, so frames in [begin, end] <=> t in [lo, hi]."""
    sample_freq = FPS // int(clips_per_second)
    secs = int(window) // int(clips_per_second)
    c = np.asarray(center_frames, dtype=np.float64).reshape(-1)
    begin = np.round(c - (float(secs) / 2.0 * FPS)).astype(np.int64)
    end = begin + secs * FPS
    lo = -((-(begin + 1)) // sample_freq) - 1
    hi = (end + 1) // sample_freq - 1
    return lo, hi


EPIC_FPS = 30   # lib/datasets/epic.py:46 / cfg.EPIC.FPS


def _ceil_div(a, b):
    return -((-a) // b)


def epic_verb_window_steps(center_frames, window, clips_per_second=1, fps=EPIC_FPS):
    """first / last bank step of sample_verb_lfb (epic.py:310-318): frames [centre - half, centre + half] with
    half = (window * FPS) // 2.  Verb banks hold the frames f with f % sample_freq == 0
    (epic.py:286-303), bank step t = frame t * sample_freq."""
    sf = int(fps) // int(clips_per_second)
    c = np.asarray(center_frames, dtype=np.int64).reshape(-1)
    half = (int(window) * int(fps)) // 2
    lo = _ceil_div(c - half, sf)
    hi = (c + half) // sf
    return lo, hi


def epic_noun_window_steps(center_frames, window, max_per_frame=10, frames_per_second=1, fps=EPIC_FPS):
    """first / last bank step of sample_noun_lfb (epic.py:338-350): secs = window / (max_per_frame * fps_lfb),
    lower = int(centre - secs / 2 * FPS)  (Python int(): truncation toward zero), upper = int(lower + secs * FPS).
    Noun banks hold one detector frame per 1 / frames_per_second seconds: step t = frame t * (FPS // fps_lfb)."""
    sf = int(fps) // int(frames_per_second)
    secs = float(window) / (int(max_per_frame) * int(frames_per_second))
    c = np.asarray(center_frames, dtype=np.float64).reshape(-1)
    lower = np.trunc(c - (secs / 2.0) * fps).astype(np.int64)
    upper = np.trunc(lower + secs * fps).astype(np.int64)
    return _ceil_div(lower, sf), upper // sf


def reference_draw_table(counts, vid, centre, clip_index, window, K, rng):
    """The slot table of `sample_lfb` (lib/datasets/ava.py:300-323) for the rows of a minibatch, drawn from `rng` in the
    reference's call order: clips in minibatch order (ava.py:205-230 appends one sampled bank per clip; rows of one clip --
    its boxes, ava_data_input.py:191-192 -- share that bank), per clip the occupied seconds of its window in ascending
    order, one `rng.choice(range(num_feat), min(num_feat, K), replace=False)` each.  counts[video][step] = features stored
    for that second (0 = `si not in in_video_lfb`); centre = keyframe second minus the bank's first second.
    -> int32 [rows][window][K], -1 = empty slot.  Pure host code (tests/test_lfb_oracle.py replays the reference loop)."""
    rows = len(vid)
    ids = np.asarray(clip_index).astype(np.int64).reshape(-1)
    if ids.size != rows or (rows and (ids[0] != 0 or np.any(np.diff(ids) < 0) or np.any(np.diff(ids) > 1))):
        raise ValueError("clip_index must be the minibatch position of every row's clip: 0, 0, 1, 1, 1, ... (got %r)" % (ids[:8],))
    table = np.full((rows, int(window), int(K)), -1, dtype=np.int32)
    prev = None
    for r in range(rows):
        if prev is None or ids[r] != ids[prev]:              # a new clip of the minibatch: its own draw
            if prev is not None and (vid[r] == vid[prev] and centre[r] == centre[prev]):
                pass                                         # (the same keyframe again: still a NEW draw, as in the reference)
            t = np.full((int(window), int(K)), -1, dtype=np.int32)
            lower = int(centre[r]) - int(window) // 2
            for j in range(int(window)):
                si = lower + j
                n = int(counts[vid[r], si]) if (0 <= vid[r] < counts.shape[0] and 0 <= si < counts.shape[1]) else 0
                if n > 0:                                    # `if si in in_video_lfb`
                    used = min(n, int(K))
                    t[j, :used] = rng.choice(range(n), used, replace=False)
            cur = t
        elif vid[r] != vid[prev] or centre[r] != centre[prev]:
            raise ValueError("rows %d and %d belong to one clip but name different keyframes" % (prev, r))
        table[r] = cur
        prev = r
    return table


class DeviceBank(object):
    """bank[video][step][slot][dim] + count[video][step] on one GPU.

    `step_base` is subtracted from the reference's time keys (AVA seconds start at 902);
    `video_ids` optionally maps the reference's video keys to dense rows."""

    def __init__(self, n_videos, n_steps, capacity, dim=2048, dtype="bf16", device="cuda:0", step_base=0,
                 video_ids=None):
        hip.lib()
        self.device = torch.device(device)
        self.code = hip.BF16 if dtype in ("bf16", torch.bfloat16) else hip.F32
        self.desc = hip.LfbDesc(int(n_videos), int(n_steps), int(capacity), int(dim), self.code, int(step_base))
        nbytes = hip.lib().vlfb_lfb_bank_bytes(C.byref(self.desc))
        if nbytes <= 0:
            raise hip.VlfbError("lfb bank: bad geometry %r" % ((n_videos, n_steps, capacity, dim),))
        self.bank = torch.zeros(int(n_videos) * int(n_steps) * int(capacity) * int(dim), device=self.device,
                                dtype=hip.TORCH_DTYPE[self.code])
        self.count = torch.zeros(int(n_videos) * int(n_steps), device=self.device, dtype=torch.int32)
        self.dropped = torch.zeros(1, device=self.device, dtype=torch.int32)
        self.step_base = int(step_base)
        self.video_row = None if video_ids is None else {v: i for i, v in enumerate(video_ids)}
        self.video_ids = None if video_ids is None else list(video_ids)

    # ---- geometry --------------------------------------------------------------------------------
    n_videos = property(lambda self: self.desc.n_videos)
    n_steps = property(lambda self: self.desc.n_steps)
    capacity = property(lambda self: self.desc.capacity)
    dim = property(lambda self: self.desc.dim)

    def _rows_of(self, videos):
        v = np.asarray(videos).astype(np.int64).reshape(-1)
        if self.video_row is None:
            return v
        return np.array([self.video_row.get(int(x), -1) if x >= 0 else -1 for x in v], dtype=np.int64)

    def _dev_i32(self, arr):
        return torch.as_tensor(np.ascontiguousarray(arr, dtype=np.int32)).to(self.device)

    # ---- construction ----------------------------------------------------------------------------
    def append(self, feats, videos, steps):
        """feats: (R, dim[,1,1,1]) device tensor (fp32 / bf16) or host array; videos/steps: (R,) reference
        keys (a negative video marks a padding row)."""
        if not torch.is_tensor(feats):
            feats = torch.as_tensor(np.asarray(feats, dtype=np.float32))
        feats = feats.to(self.device).reshape(feats.shape[0], -1).contiguous()
        assert feats.shape[1] == self.dim, "append: feature width %d, bank dim %d" % (feats.shape[1], self.dim)
        rows = feats.shape[0]
        keys = np.stack([self._rows_of(videos), np.asarray(steps).astype(np.int64).reshape(-1) - self.step_base], axis=1)
        assert keys.shape == (rows, 2)
        kd = self._dev_i32(keys)
        hip.call("vlfb_lfb_append", C.byref(self.desc), hip.ptr(self.bank), hip.ptr(self.count), hip.ptr(feats),
                 hip.dtype_code(feats.dtype), hip.ptr(kd), rows, hip.ptr(self.dropped))
        torch.cuda.current_stream().synchronize()     # kd / feats may be temporaries

    def append_ava(self, box_pooled, metadata):
        """one inference iteration of construct_ava_lfb (lfb_loader.py:79-112): metadata rows are
        [video_id, sec, ...] as floats"""
        md = np.asarray(metadata, dtype=np.float64)
        self.append(box_pooled, np.round(md[:, 0]).astype(np.int64), np.round(md[:, 1]).astype(np.int64))

    def append_frames(self, pool5, frame_keys, sample_freq):
        """one inference iteration of construct_frame_level_lfb (lfb_loader.py:49-76): frame_keys are
        (video, frame) pairs with (frame + 1) % sample_freq == 0 (charades.py:233-248); rows of
        `pool5` beyond len(frame_keys) are padding"""
        fk = np.asarray(frame_keys, dtype=np.int64).reshape(-1, 2)
        rows = pool5.shape[0]
        vids = np.full(rows, -1, dtype=np.int64)
        steps = np.zeros(rows, dtype=np.int64)
        n = min(rows, fk.shape[0])
        assert np.all((fk[:n, 1] + 1) % sample_freq == 0), "frame keys must be LFB frames"
        vids[:n] = fk[:n, 0]
        steps[:n] = (fk[:n, 1] + 1) // sample_freq - 1
        self.append(pool5, vids, steps + self.step_base)

    def check_no_drops(self):
        n = int(self.dropped.item())
        if n:
            raise hip.VlfbError("lfb bank: %d features did not fit (capacity %d per step, or key out of range)"
                                % (n, self.capacity))

    def counts(self):
        return self.count.view(self.n_videos, self.n_steps).cpu().numpy()

    # ---- sampling --------------------------------------------------------------------------------
    def _out(self, out, shape, dtype):
        if out is None:
            return torch.empty(shape, device=self.device, dtype=dtype or hip.TORCH_DTYPE[self.code])
        assert out.is_contiguous() and out.numel() == int(np.prod(shape)), "sample: output tensor has the wrong size"
        return out

    def sample_window(self, videos, secs, sample_ids, window, max_per_step, seed, out=None, out_dtype=None):
        """AVA (ava.py:300-323): -> (R, window*max_per_step, dim).  `sample_ids` name the random draw
        (e.g. iteration * batch + clip); RoIs of one clip pass the same id and get the same sample."""
        rows = len(videos)
        q = np.stack([self._rows_of(videos), np.asarray(secs).astype(np.int64).reshape(-1) - self.step_base,
                      np.asarray(sample_ids).astype(np.int64).reshape(-1) & 0x7FFFFFFF], axis=1)
        qd = self._dev_i32(q)
        out = self._out(out, (rows, window * max_per_step, self.dim), out_dtype)
        hip.call("vlfb_lfb_sample_window", C.byref(self.desc), hip.ptr(self.bank), hip.ptr(self.count), hip.ptr(qd),
                 rows, int(window), int(max_per_step), C.c_uint64(int(seed) & 0xFFFFFFFFFFFFFFFF), hip.ptr(out),
                 hip.dtype_code(out.dtype))
        torch.cuda.current_stream().synchronize()
        return out

    def sample_window_reference_draw(self, videos, secs, clip_index, window, max_per_step, rng=None, out=None, out_dtype=None):
        """AVA (ava.py:300-323) with the reference's OWN random stream: the host makes exactly the calls `sample_lfb` makes
        (reference_draw_table below) and the device gathers (vlfb_lfb_gather_slots).  `clip_index[r]` = the position of row r's
        CLIP inside the minibatch (0, 0, 1, 1, 1, ...: the batch-index column the data layer writes into `proposals`,
        ava_data_input.py:175-192) -- NOT a keyframe id: the reference calls sample_lfb once per clip of the minibatch
        (ava.py:230), so a keyframe that was drawn twice into one minibatch makes two draws.  With `rng = np.random` seeded as
        the reference seeds it (np.random.seed(cfg.RNG_SEED)) and the caller's other np.random calls interleaved as the
        reference interleaves them, the sampled banks are the reference's, element for element.  `sample_window` (a
        counter-based key per draw, no host work, no sync) has the same DISTRIBUTION but another stream.  Needs the step
        counts on the host: one small D2H copy per call."""
        rng = np.random if rng is None else rng
        rows = len(videos)
        vid = self._rows_of(videos)
        centre = np.asarray(secs).astype(np.int64).reshape(-1) - self.step_base
        K = int(max_per_step)
        table = reference_draw_table(self.counts(), vid, centre, clip_index, int(window), K, rng)
        qd = self._dev_i32(np.stack([vid, centre], axis=1))
        td = self._dev_i32(table.reshape(-1))
        out = self._out(out, (rows, int(window) * K, self.dim), out_dtype)
        hip.call("vlfb_lfb_gather_slots", C.byref(self.desc), hip.ptr(self.bank), hip.ptr(self.count), hip.ptr(qd), hip.ptr(td),
                 rows, int(window), K, hip.ptr(out), hip.dtype_code(out.dtype))
        torch.cuda.current_stream().synchronize()
        return out

    def sample_frames(self, videos, center_frames, window, clips_per_second, out=None, out_dtype=None):
        """Charades (charades.py:251-276): -> (N, window, dim), the first `window` bank frames inside
        [begin, end] around each clip centre, packed to the front"""
        rows = len(videos)
        lo, hi = frame_window_steps(center_frames, window, clips_per_second)
        q = np.stack([self._rows_of(videos), lo, hi], axis=1)
        qd = self._dev_i32(q)
        out = self._out(out, (rows, int(window), self.dim), out_dtype)
        hip.call("vlfb_lfb_sample_compact", C.byref(self.desc), hip.ptr(self.bank), hip.ptr(self.count), hip.ptr(qd),
                 rows, int(window), hip.ptr(out), hip.dtype_code(out.dtype))
        torch.cuda.current_stream().synchronize()
        return out

    def _sample_packed(self, videos, lo, hi, window, max_per_step, out, out_dtype):
        rows = len(videos)
        qd = self._dev_i32(np.stack([self._rows_of(videos), lo, hi], axis=1))
        out = self._out(out, (rows, int(window), self.dim), out_dtype)
        hip.call("vlfb_lfb_sample_packed", C.byref(self.desc), hip.ptr(self.bank), hip.ptr(self.count), hip.ptr(qd),
                 rows, int(window), int(max_per_step), hip.ptr(out), hip.dtype_code(out.dtype))
        torch.cuda.current_stream().synchronize()
        return out

    def sample_epic_verb(self, videos, center_frames, window, clips_per_second=1, out=None, out_dtype=None):
        """EPIC-Kitchens verb model (epic.py:310-331): -> (N, window, dim), the first `window` bank clips whose
        centre frame lies within +-(window * FPS) // 2 frames of the clip centre, zero padded"""
        lo, hi = epic_verb_window_steps(center_frames, window, clips_per_second)
        return self._sample_packed(videos, lo, hi, window, 1, out, out_dtype)

    def sample_epic_noun(self, videos, center_frames, window, max_per_frame=10, frames_per_second=1, out=None,
                         out_dtype=None):
        """EPIC-Kitchens noun model (epic.py:338-374): detector features, at most `max_per_frame` per bank frame
        in stored order, frames in time order, truncated to `window` rows, zero padded"""
        lo, hi = epic_noun_window_steps(center_frames, window, max_per_frame, frames_per_second)
        return self._sample_packed(videos, lo, hi, window, max_per_frame, out, out_dtype)

    @classmethod
    def from_epic(cls, lfb, noun, sample_freq=EPIC_FPS, capacity=None, dtype="bf16", device="cuda:0"):
        """`lfb` as the EPIC datasets hold it: verb {video: {frame: feat}} (frames % sample_freq == 0), noun
        {video: {frame: (n, dim) array or []}} (epic.py:191-199)"""
        vids = sorted(lfb)
        frames = [f for v in lfb.values() for f in v]
        assert all(f % sample_freq == 0 for f in frames), "EPIC bank frames must be multiples of the sampling period"
        n_steps = max(frames) // sample_freq + 1 if frames else 1
        feats = [np.asarray(x, dtype=np.float32).reshape(-1, np.shape(x)[-1]) for v in lfb.values() for x in v.values()
                 if np.size(x)]
        dim = feats[0].shape[1]
        cap = capacity or (max(f.shape[0] for f in feats) if noun else 1)
        bank = cls(len(vids), n_steps, cap, dim, dtype, device, 0, vids)
        for v in vids:
            rows, ks = [], []
            for f in sorted(lfb[v]):
                x = lfb[v][f]
                if not np.size(x):
                    continue
                x = np.asarray(x, dtype=np.float32).reshape(-1, dim)
                rows.append(x[:cap])
                ks += [f // sample_freq] * min(cap, x.shape[0])
            if rows:
                bank.append(torch.as_tensor(np.concatenate(rows)), [v] * len(ks), ks)
        bank.check_no_drops()
        return bank

    # ---- interchange with the reference's pickles -----------------------------------------------
    @classmethod
    def from_reference(cls, lfb, capacity=None, dtype="bf16", device="cuda:0", frame_level=False, sample_freq=None):
        """`lfb` as tools/lfb_loader.py writes it: {video: {sec: [feat, ...]}} or, frame_level,
        {video: {frame: feat}}"""
        vids = sorted(lfb)
        if frame_level:
            assert sample_freq, "frame-level banks need sample_freq (FPS // LFB_CLIPS_PER_SECOND)"
            n_steps = max([(max(f) + 1) // sample_freq for f in lfb.values() if len(f)] + [1])
            first = next(iter(next(v for v in lfb.values() if len(v)).values()))
            bank = cls(len(vids), n_steps, 1, int(np.size(first)), dtype, device, 0, vids)
            for v in vids:
                frames = sorted(lfb[v])
                if frames:
                    bank.append_frames(torch.as_tensor(np.stack([np.ravel(lfb[v][f]) for f in frames]).astype(np.float32)),
                                       [(v, f) for f in frames], sample_freq)
            return bank
        secs = [s for v in lfb.values() for s in v]
        base, n_steps = (min(secs), max(secs) - min(secs) + 1) if secs else (0, 1)
        cap = capacity or max([len(l) for v in lfb.values() for l in v.values()] + [1])
        first = next(l[0] for v in lfb.values() for l in v.values() if len(l))
        bank = cls(len(vids), n_steps, cap, int(np.size(first)), dtype, device, base, vids)
        for v in vids:
            feats, ks = [], []
            for s in sorted(lfb[v]):
                for f in lfb[v][s]:
                    feats.append(np.ravel(f))
                    ks.append(s)
            if feats:
                bank.append(torch.as_tensor(np.stack(feats).astype(np.float32)), [v] * len(ks), ks)
        bank.check_no_drops()
        return bank

    def to_reference(self, frame_level=False, sample_freq=None):
        cnt = self.counts()
        data = self.bank.view(self.n_videos, self.n_steps, self.capacity, self.dim).float().cpu().numpy()
        out = {}
        for vi in range(self.n_videos):
            key = self.video_ids[vi] if self.video_ids is not None else vi
            out[key] = {}
            for s in np.nonzero(cnt[vi])[0]:
                if frame_level:
                    out[key][int(sample_freq * (s + 1) - 1)] = data[vi, s, 0].copy()
                else:
                    out[key][int(s + self.step_base)] = [data[vi, s, k].copy() for k in range(cnt[vi, s])]
        return out
