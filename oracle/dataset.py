"""TEST INFRASTRUCTURE ONLY -- numpy restatement of the reference's training-item and collate arithmetic
(datasets.py:205-224 resampling, 251-368 `DatasetPickle.__getitem__`, 424-503 `collate_fn`).  Pinned against the
reference's own class (built from its source's AST in this container) by tests/golden/g7_dataset.npz.
Only tests/ may import this module."""
import numpy as np


def resample(x, original_fps, coef_fps):
    """datasets.py:205-224 (scipy interp1d, linear, normalised time axis) via np.interp per column."""
    x = np.asarray(x, dtype=np.float64)
    n = x.shape[0]
    n_new = int(round(n / original_fps * coef_fps))
    xs, xn = np.linspace(0, 1, num=n), np.linspace(0, 1, num=n_new)
    return np.stack([np.interp(xn, xs, x[:, c]) for c in range(x.shape[1])], axis=1)


def get_item(clip, coef_stats, rng, audio_unit=640.0, n_motions=100, clip_len=100, random_crop=True):
    """-> ([audio_0, audio_1], [motion_0, motion_1], (mean, std)); `clip` holds 25 fps tracks."""
    audio = np.asarray(clip["audio"], dtype=np.float32)
    e, h = np.asarray(clip["expression_code"]), np.asarray(clip["head_orientation"])
    mean, std = audio.mean(), audio.std()
    audio = (audio - mean) / (std + 1e-5)
    goal = int(n_motions * 2.1)
    cur = e.shape[0]
    s1 = 0
    if random_crop and cur > goal:
        s1 = rng.randint(0, cur - goal + 1)
    elif random_crop and cur < goal:
        pad = goal - cur
        front = rng.randint(0, pad)
        back = pad - front
        e = np.pad(e, ((front, back), (0, 0)))
        h = np.pad(h, ((front, back), (0, 0)))
        audio = np.pad(audio, (int(round(front * audio_unit)), int(round(back * audio_unit))))
        need = int(round(goal * audio_unit))
        if audio.shape[0] < need:
            audio = np.pad(audio, (0, need - audio.shape[0]))
    elif not random_crop:
        e = np.pad(e, ((0, int(round(goal - cur))), (0, 0)))
        h = np.pad(h, ((0, int(round(goal - cur))), (0, 0)))
        audio = np.pad(audio, (0, int(round(goal * audio_unit)) - audio.shape[0]))
    out_a, out_m = [], []
    for w in range(2):
        a, b = s1 + w * clip_len, s1 + (w + 1) * clip_len
        ew, hw = e[a:b].astype(np.float32), h[a:b].astype(np.float32)
        if coef_stats is not None:
            ew = (ew - coef_stats["exp_mean"]) / (coef_stats["exp_std"] + np.float32(1e-9))
            hw = (hw - coef_stats["pose_mean"]) / (coef_stats["pose_std"] + np.float32(1e-9))
        out_m.append(np.concatenate([ew, hw], axis=-1).astype(np.float32))
        out_a.append(audio[int(a * audio_unit):int(b * audio_unit)].astype(np.float32))
    return out_a, out_m, (mean, std)


def collate(items, target=64000):
    """datasets.py:440-503: pad / trim every audio window to 64000 samples and stack."""
    fix = lambda a: np.pad(a, (0, target - a.shape[0])) if a.shape[0] < target else a[:target]
    audio = [np.stack([fix(it[0][w]) for it in items]) for w in range(2)]
    motion = [np.stack([it[1][w] for it in items]) for w in range(2)]
    stats = (np.float32(np.mean(np.asarray([it[2][0] for it in items], np.float32))),
             np.float32(np.mean(np.asarray([it[2][1] for it in items], np.float32))))
    return audio, motion, stats
