"""Shared test helpers: synthetic state dicts and inputs (same recipes as tests/golden/make_goldens.py)."""
import functools

import numpy as np

from msmd_amd import shapes, synth
from msmd_amd.config import default_args


@functools.lru_cache(maxsize=4)
def msmd_state_dict(audio_model="wav2vec2", n_encoder_layers=12, **kw):
    args = default_args(audio_model=audio_model, **kw)
    return synth.fill_state_dict(shapes.msmd_shapes(args, n_encoder_layers)), args


@functools.lru_cache(maxsize=2)
def style_state_dict():
    args = default_args()
    return synth.fill_state_dict(shapes.style_encoder_shapes(args)), args


def denoiser_inputs(B, args, tag="dn"):
    d = args.feature_dim
    return dict(
        motion=synth.normalish(f"{tag}/motion", (B, 100, 67)),
        audio_feat=synth.normalish(f"{tag}/audio_feat", (B, 100, d)),
        shape=(0.3 * synth.normalish(f"{tag}/shape", (B, 100))).astype(np.float32),
        style=synth.normalish(f"{tag}/style", (B, args.d_style)),
        prev_motion=synth.normalish(f"{tag}/prev_motion", (B, 10, 67)),
        prev_audio=synth.normalish(f"{tag}/prev_audio", (B, 10, d)),
        indicator=np.concatenate([np.ones((B, 80), np.float32), np.zeros((B, 20), np.float32)], axis=1),
    )


def flame_inputs(B, tag="flame"):
    return dict(shape=(0.5 * synth.normalish(f"{tag}/shape", (B, 100))).astype(np.float32),
                exp=(0.5 * synth.normalish(f"{tag}/exp", (B, 50))).astype(np.float32),
                pose=(0.4 * synth.normalish(f"{tag}/pose", (B, 6))).astype(np.float32))


def maxabs(a, b):
    return float(np.max(np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64))))


@functools.lru_cache(maxsize=1)
def hubert_large_state_dict(n_layers=2):
    """Encoder-only synthetic state dict of the HuBERT-large ARCHITECTURE (keys prefixed audio_encoder.)."""
    shp = shapes.audio_encoder_shapes(n_layers, 1024, 4096, feat_extract_norm="layer", conv_bias=True)
    return synth.fill_state_dict({"audio_encoder." + k: v for k, v in shp.items()})


def ingest_inputs(n, tag="ingest"):
    """Synthetic style clip + corpus statistics: the recipe of tests/golden/make_goldens.py::ingest_inputs (g3_ingest)."""
    e = (synth.normalish(f"{tag}/{n}/exp", (n, 50)) * 1.5 + 0.2).astype(np.float32)
    h = (synth.normalish(f"{tag}/{n}/head", (n, 3)) * 0.3).astype(np.float32)
    st = {"exp_mean": synth.normalish(f"{tag}/exp_mean", (50,)) * 0.1, "exp_std": np.abs(synth.normalish(f"{tag}/exp_std", (50,))) + 0.5,
          "pose_mean": synth.normalish(f"{tag}/pose_mean", (3,)) * 0.1, "pose_std": np.abs(synth.normalish(f"{tag}/pose_std", (3,))) + 0.5}
    return e, h, {k: v.astype(np.float32) for k, v in st.items()}


def check_ingestion_against_reference(tmp_path, device):
    """inference.query_for_motion_coeff on pickle files, as its caller feeds it, against the outputs of the REFERENCE's own
    function on the same files (tests/golden/g3_ingest.npz: reference inference.py:109-183 built from its AST)."""
    import pickle

    import torch

    from conftest import load_golden
    from msmd_amd.inference import query_for_motion_coeff
    from types import SimpleNamespace
    g = load_golden("g3_ingest")
    for case in g["cases"]:
        n, fps, kind = str(case).split(",")
        n, fps = int(n), int(fps)
        e, h, st = ingest_inputs(n)
        paths = {k: tmp_path / f"{k}_{n}.pkl" for k in ("stats", "exp", "head")}
        for k, obj in (("stats", {a: torch.from_numpy(b) for a, b in st.items()}), ("exp", torch.from_numpy(e)),
                       ("head", torch.from_numpy(h) if kind == "tensor" else h)):
            with open(paths[k], "wb") as f:
                pickle.dump(obj, f)
        motion, shape = query_for_motion_coeff(SimpleNamespace(coef_dict_path=str(paths["stats"])), str(paths["exp"]),
                                               str(paths["head"]), device=device, original_fps=fps, target_fps=25)
        want = g[f"motion_{n}_{fps}"]
        assert motion.device.type == torch.device(device).type and motion.dtype == torch.float32
        assert tuple(motion.shape) == want.shape and tuple(shape.shape) == g[f"shape_{n}_{fps}"].shape == (1, 100)
        # same float64 arithmetic in a different association (interp1d's slope form): a few ulp of fp32 after the final cast
        assert float(np.abs(motion.cpu().numpy() - want).max()) <= 1e-6, (n, fps)
        assert float(shape.abs().sum()) == 0.0
