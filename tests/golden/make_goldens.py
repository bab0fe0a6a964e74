#!/usr/bin/env python3
"""Generate the committed golden vectors by running the REFERENCE itself.

Runs only in the build container, where /root/reference is mounted read-only:
it imports the reference's modules with the 3-line shim of SURVEY.md Appendix C
(random-init HF configs instead of from_pretrained, enc_dec_mask on CPU, an
explicit args namespace), fills every parameter with the closed-form synthetic
weights of ``msmd_amd.synth`` and records inputs/outputs as small ``.npz``
files next to this script.  No reference source is copied: the fixtures are
data (inputs, expected outputs).  ``infer_coeffs`` lives in a script that
cannot be imported here (needs cv2/librosa); its function object is built from
the reference file's AST at generation time only.

Usage:  PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_goldens.py [names...]
"""
from __future__ import annotations

import argparse
import ast
import math
import os
import pickle
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True
REF = "/root/reference"
sys.path.insert(0, REF)

import torch  # noqa: E402
import transformers  # noqa: E402

from msmd_amd import synth, shapes  # noqa: E402
from msmd_amd.config import default_args  # noqa: E402

torch.set_grad_enabled(False)
VERSIONS = f"torch={torch.__version__};transformers={transformers.__version__};numpy={np.__version__}"


# --------------------------------------------------------------------------- shim
def import_reference():
    import utils.wav2vec2 as w2
    import utils.hubert as hb
    import utils.model_common as mc
    from transformers import Wav2Vec2Config, HubertConfig
    w2.Wav2Vec2Model.from_pretrained = classmethod(
        lambda cls, name, **kw: cls(Wav2Vec2Config(attn_implementation="eager")))
    hb.HubertModel.from_pretrained = classmethod(
        lambda cls, name, **kw: cls(HubertConfig(attn_implementation="eager")))
    import model as M
    import style_encoder as SE
    orig = mc.enc_dec_mask
    M.enc_dec_mask = lambda T, S, fw=2, ex=0, device="cpu": orig(T, S, fw, ex, device="cpu")
    return M, SE, mc


def ref_args(**kw):
    a = default_args(**kw)
    return a


def build_ref_model(M, audio_model="wav2vec2", **kw):
    args = ref_args(audio_model=audio_model, **kw)
    model = M.get_diffusion_model(args, device="cpu").eval()
    synth.load_synthetic(model)
    return model, args


def save(name, **arrays):
    path = os.path.join(HERE, name + ".npz")
    arrays["_versions"] = np.array(VERSIONS)
    np.savez_compressed(path, **{k: np.asarray(v) for k, v in arrays.items()})
    print(f"wrote {path}  ({os.path.getsize(path) / 1024:.1f} KiB)")


def t(x):
    return torch.from_numpy(np.ascontiguousarray(x))


# --------------------------------------------------------------------------- G1 index maths
def g1_index(M, SE, mc):
    out = {}
    for L in (31999, 32000, 32001, 32081, 32320, 64000, 64001, 64079, 64080, 64081, 160000):
        x = torch.arange(L, dtype=torch.float32)[None]
        y = mc.pad_audio(x)[0].numpy()
        out[f"pad_len_{L}"] = np.int64(y.shape[0])
        out[f"pad_head_{L}"] = y[:48].astype(np.int64)
        out[f"pad_tail_{L}"] = y[-48:].astype(np.int64)
    # conv length chain from the real HF module
    import utils.wav2vec2 as w2
    enc = w2.Wav2Vec2Model.from_pretrained("x")
    for S in (32080, 64080, 160080):
        out[f"conv_T_{S}"] = np.int64(enc._get_feat_extract_output_lengths(torch.tensor(S)).item())
    # crop + interpolate tables, obtained by pushing index ramps through the reference's function
    for fps, frame_num, T50 in ((25, 200, 200), (30, 200, 400), (25, 500, 500), (25, 100, 100), (30, 120, 250)):
        crop = round(frame_num * 50 / fps)
        x = torch.arange(T50, dtype=torch.float32)[None, None, :crop]
        y = w2.linear_interpolation(x, 50, fps, output_len=frame_num)[0, 0].numpy()
        out[f"crop_{fps}_{frame_num}"] = np.int64(crop)
        out[f"interp_{fps}_{frame_num}_{T50}"] = y
    import torch.nn.functional as F
    y = F.interpolate(torch.arange(200, dtype=torch.float32)[None, None], size=100, align_corners=False, mode="linear")
    out["interp_200_to_100"] = y[0, 0].numpy()
    # infer_coeffs window plan
    for S in (32000, 64000, 100000, 200001, 64001, 63999):
        clip_len = int(S / 16000 * 25)
        n_sub = 1 if clip_len <= 100 else math.ceil(clip_len / 100)
        n_pad = round(640.0 * 100) * n_sub - S
        out[f"plan_{S}"] = np.array([clip_len, n_sub, n_pad, math.ceil(n_pad / 640.0)], dtype=np.int64)
    save("g1_index", **out)


# --------------------------------------------------------------------------- G2 schedule
def g2_schedule(M, SE, mc):
    out = {}
    for T in (5, 500):
        for mode in ("linear", "quadratic", "sigmoid", "cosine"):
            s = M.DiffusionSchedule(T, mode)
            for k in ("betas", "alphas", "alpha_bars", "sigmas_flex", "sigmas_inflex"):
                out[f"{mode}_{T}_{k}"] = getattr(s, k).numpy()
    out["enc_dec_mask_110_1_0"] = mc.enc_dec_mask(110, 110, 1, 0, device="cpu").numpy()
    out["enc_dec_mask_110_1_1"] = mc.enc_dec_mask(110, 110, 1, 1, device="cpu").numpy()
    pe = mc.PositionalEncoding(512, max_len=501)
    out["pe_512_501_rows"] = pe.pe[0, [0, 1, 7, 250, 500]].numpy()
    save("g2_schedule", **out)


# --------------------------------------------------------------------------- G3 network blocks
def g3_audio(M, SE, mc):
    for am in ("wav2vec2", "hubert"):
        model, args = build_ref_model(M, am)
        enc = model.audio_encoder
        B = 2
        audio = synth.audio_clips(B, 64000)
        grabs = {}

        def hook(name):
            def fn(mod, inp, outp):
                o = outp[0] if isinstance(outp, tuple) else outp
                grabs[name] = o.detach().numpy().copy()
            return fn
        hs = [enc.feature_extractor.conv_layers[0].register_forward_hook(hook("conv0")),
              enc.feature_extractor.register_forward_hook(hook("conv")),
              enc.feature_projection.register_forward_hook(hook("proj")),
              enc.encoder.pos_conv_embed.register_forward_hook(hook("posconv")),
              enc.encoder.layers[0].register_forward_hook(hook("layer0")),
              enc.encoder.layers[11].register_forward_hook(hook("layer11"))]
        feat = model.extract_audio_feature(t(audio)).numpy()
        feat768 = model.extract_audio_768_feature(t(audio)).numpy()
        for h in hs:
            h.remove()
        out = dict(
            conv0=grabs["conv0"].transpose(0, 2, 1)[:, ::61, ::7],      # (B, 12815, 512) channels-last, subsampled
            conv=grabs["conv"].transpose(0, 2, 1)[:, ::3, ::5],        # (B, 200, 512)
            proj=grabs["proj"][:, ::3, ::5],
            posconv=grabs["posconv"][:, ::3, ::5],                      # GELU(conv)[:, :-1] (B,200,768)
            layer0=grabs["layer0"][:, ::3, ::5],
            layer11=grabs["layer11"][:, ::3, ::5],
            feat768=feat768[:, ::2, ::3],
            feat=feat,
        )
        # 2 s clip at 30 fps exercises the non-trivial crop/interp path of the wrapper
        a2 = synth.audio_clips(1, 32000, tag="audio30")
        y = enc(mc.pad_audio(t(a2)), 30, frame_num=60).last_hidden_state.numpy()
        out["hidden_fps30_60"] = y
        save(f"g3_audio_{am}", **out)
        if am == "wav2vec2":
            keys = list(model.state_dict().keys())
            shp = {k: tuple(v.shape) for k, v in model.state_dict().items()}
            mine = shapes.msmd_shapes(args)
            for k, v in mine.items():
                kk = k.replace("weight_g", "parametrizations.weight.original0").replace(
                    "weight_v", "parametrizations.weight.original1")
                assert shp[kk] == tuple(v), (k, shp[kk], v)
            extra = [k for k in keys if synth.canonical_name(k) not in mine and not synth.is_computed_buffer(k)]
            assert not extra, extra
            n_params = sum(int(np.prod(v)) for v in mine.values())
            save("g0_keys", keys=np.array(keys), n_params=np.int64(n_params),
                 style_keys=np.array(list(SE.get_style_encoder(args, "vae2").state_dict().keys())))


def g2_lr_schedule(M, SE, mc):
    """Learning-rate sequences of the reference's scheduler set-up (training_script.py:571-581) stepped as its training
    loop does (l.222-224), using the reference's own utils/scheduler.py class."""
    from utils.scheduler import GradualWarmupScheduler
    out = {}
    cases = [("Warmup", 1e-3, 4, 12, 0.1, 10), ("WarmupThenDecay", 1e-3, 4, 12, 0.1, 20),
             ("WarmupThenDecay", 2e-5, 50, 400, 0.1, 450), ("Warmup", 2e-5, 5000, 0, 0.1, 5005)]
    out["cases"] = np.array([(c[0], *map(str, c[1:])) for c in cases])
    for i, (kind, lr, warm, cos_max, ratio, n) in enumerate(cases):
        opt = torch.optim.Adam([torch.nn.Parameter(torch.zeros(1))], lr=lr)
        if kind == "Warmup":
            sch = GradualWarmupScheduler(opt, 1, warm)
        else:
            after = torch.optim.lr_scheduler.CosineAnnealingLR(opt, cos_max - warm, lr * ratio)
            sch = GradualWarmupScheduler(opt, 1, warm, after)
        seq = []
        for it in range(n):
            seq.append(opt.param_groups[0]["lr"])
            opt.step()
            if kind != "WarmupThenDecay" or it < cos_max:
                sch.step()
        out[f"lr_{i}"] = np.array(seq, dtype=np.float64)
    save("g2_lr_schedule", **out)


def dataset_raw_clips():
    """Synthetic decoded corpus (30 fps tracks, 64-d expression codes + 3-d head orientation, 16 kHz audio): clips
    longer than / equal to / shorter than the 210-frame item length after the 30 -> 25 fps resampling."""
    raw = {}
    for name, n30 in (("long_a", 400), ("long_b", 301), ("exact", 252), ("short_a", 200), ("short_b", 131), ("tiny", 40)):
        n25 = int(round(n30 / 30 * 25))
        S = int(n25 * 640) + {"long_a": 37, "long_b": -211, "exact": 0, "short_a": 5, "short_b": -400, "tiny": 123}[name]
        raw[name] = {"audio": (0.3 * synth.normalish(f"ds/{name}/audio", (S,)) + 0.05).astype(np.float32),
                     "expression_code": synth.normalish(f"ds/{name}/exp", (n30, 64)).astype(np.float64) * 1.5 + 0.2,
                     "head_orientation": synth.normalish(f"ds/{name}/head", (n30, 3)).astype(np.float64) * 0.3}
    return raw


def _load_dataset_class():
    """Build the reference's DatasetPickle / incremental_mean_and_std from its file's AST (datasets.py imports
    torchaudio / librosa / cv2, absent here): this container only."""
    import pickle as _pickle
    from scipy.interpolate import interp1d
    from torch.utils import data
    src = open(os.path.join(REF, "datasets.py")).read()
    tree = ast.parse(src)
    keep = [n for n in tree.body if (isinstance(n, ast.ClassDef) and n.name == "DatasetPickle")
            or (isinstance(n, ast.FunctionDef) and n.name == "incremental_mean_and_std")]
    ns = {"np": np, "torch": torch, "data": data, "interp1d": interp1d, "pickle": _pickle, "tqdm": (lambda x: x)}
    exec(compile(ast.Module(body=keep, type_ignores=[]), "<reference datasets.py>", "exec"), ns)
    return ns["DatasetPickle"], ns["incremental_mean_and_std"]


def g7_dataset(M, SE, mc):
    """Items and collated batches of the reference's DatasetPickle on a synthetic corpus, numpy seeded."""
    DatasetPickle, _ = _load_dataset_class()
    raw = dataset_raw_clips()
    names = list(raw)
    tmp = tempfile.mkdtemp()
    split = os.path.join(tmp, "split.txt")
    open(split, "w").write("\n".join(names) + "\n")
    stats_file = os.path.join(tmp, "stats.npz")
    st = {"exp_mean": synth.normalish("ds/exp_mean", (64,)) * 0.1, "exp_std": np.abs(synth.normalish("ds/exp_std", (64,))) + 0.5,
          "pose_mean": synth.normalish("ds/pose_mean", (3,)) * 0.1, "pose_std": np.abs(synth.normalish("ds/pose_std", (3,))) + 0.5}
    np.savez(stats_file, **{k: v.astype(np.float32) for k, v in st.items()})
    out = {"names": np.array(names)}
    for mode, rc in (("crop", True), ("nocrop", False)):
        ds = DatasetPickle("unused", split, coef_stats_file=stats_file, original_fps=30, coef_fps=25, n_motions=100,
                           clip_len=100, pre_loaded_raw_dataset={k: dict(v) for k, v in raw.items()}, celebv_text=False,
                           random_crop=rc)
        order = [0, 3, 1, 4, 2, 5, 0, 4] if rc else [2, 3, 4, 5]   # the reference crashes on clips > 210 frames without random_crop
        np.random.seed(7)
        items = [ds[i] for i in order]
        batch = DatasetPickle.get_collate_fn(SE=False)(items)
        out[f"{mode}_order"] = np.array(order)
        out[f"{mode}_audio0"] = batch[0][0].numpy()[:, ::13]
        out[f"{mode}_audio1"] = batch[0][1].numpy()[:, ::13]
        out[f"{mode}_audio0_head"] = batch[0][0].numpy()[:, :64]
        out[f"{mode}_audio1_tail"] = batch[0][1].numpy()[:, -4000:]
        out[f"{mode}_motion0"] = batch[1][0]["motion"].numpy()
        out[f"{mode}_motion1"] = batch[1][1]["motion"].numpy()
        out[f"{mode}_stats"] = np.array([float(batch[2][0]), float(batch[2][1])])
        out[f"{mode}_item_audio_len"] = np.array([[it[0][0].shape[0], it[0][1].shape[0]] for it in items])
    # corpus statistics as the reference computes them when no stats file is given (seeded crops, no normalisation)
    np.random.seed(11)
    ds = DatasetPickle("unused", split, coef_stats_file=None, original_fps=30, coef_fps=25, n_motions=100, clip_len=100,
                       pre_loaded_raw_dataset={k: dict(v) for k, v in raw.items()}, celebv_text=False, random_crop=True)
    for k, v in ds.coef_stats.items():
        out["stats_" + k] = v.numpy()
    save("g7_dataset", **out)


def g3_audio_large(M, SE, mc):
    """HuBERT-large ARCHITECTURE (feat_extract_norm='layer' + conv biases, stable-layer-norm encoder, 1024 wide, 16
    heads) through the reference's own wrapper class utils/hubert.py:9-51, 2 transformer layers, synthetic weights;
    BASELINE.json configs[3] (10 s clip -> 500 frames at 50 fps, 250 at 25 fps x 2)."""
    import utils.hubert as hb
    from transformers import HubertConfig
    cfg = HubertConfig(hidden_size=1024, num_hidden_layers=2, num_attention_heads=16, intermediate_size=4096,
                       feat_extract_norm="layer", do_stable_layer_norm=True, conv_bias=True,
                       attn_implementation="eager")
    enc = hb.HubertModel(cfg).eval()
    synth.load_synthetic(enc, prefix="audio_encoder.")
    keys = {k: tuple(v.shape) for k, v in enc.state_dict().items()}
    mine = shapes.audio_encoder_shapes(2, 1024, 4096, feat_extract_norm="layer", conv_bias=True)
    for k, v in mine.items():
        kk = k.replace("weight_g", "parametrizations.weight.original0").replace("weight_v", "parametrizations.weight.original1")
        assert keys[kk] == tuple(v), (k, keys[kk], v)
    assert len([k for k in keys if not synth.is_computed_buffer("audio_encoder." + k)]) == len(mine)
    grabs = {}

    def hook(name):
        def fn(mod, inp, outp):
            o = outp[0] if isinstance(outp, tuple) else outp
            grabs[name] = o.detach().numpy().copy()
        return fn
    hs = [enc.feature_extractor.conv_layers[0].register_forward_hook(hook("conv0")),
          enc.feature_extractor.register_forward_hook(hook("conv")),
          enc.feature_projection.register_forward_hook(hook("proj")),
          enc.encoder.layers[0].register_forward_hook(hook("layer0")),
          enc.encoder.layers[1].register_forward_hook(hook("layer1"))]
    out = {}
    with torch.no_grad():
        a10 = synth.audio_clips(1, 160000, tag="audio10s")
        y = enc(mc.pad_audio(t(a10)), 25, frame_num=500).last_hidden_state.numpy()      # (1, 500, 1024)
        out["hidden_10s"] = y[:, ::2, ::3]
        out["conv0_10s"] = grabs["conv0"].transpose(0, 2, 1)[:, ::127, ::7]
        out["conv_10s"] = grabs["conv"].transpose(0, 2, 1)[:, ::5, ::5]
        out["proj_10s"] = grabs["proj"][:, ::5, ::5]
        out["layer0_10s"] = grabs["layer0"][:, ::5, ::5]
        out["layer1_10s"] = grabs["layer1"][:, ::5, ::5]
        a4 = synth.audio_clips(2, 64000)
        out["hidden_4s"] = enc(mc.pad_audio(t(a4)), 25, frame_num=200).last_hidden_state.numpy()[:, ::2, ::3]
        a2 = synth.audio_clips(1, 32000, tag="audio30")
        out["hidden_fps30_60"] = enc(mc.pad_audio(t(a2)), 30, frame_num=60).last_hidden_state.numpy()
    for h in hs:
        h.remove()
    save("g3_audio_hubert_large", **out)


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


def g3_denoiser(M, SE, mc):
    model, args = build_ref_model(M)
    B = 2
    x = denoiser_inputs(B, args)
    step = torch.tensor([7, 433])
    person = torch.cat([t(x["shape"])[:, None], t(x["style"])[:, None]], dim=-1)
    out = {}
    for width in (1, 2):
        if width != 1:
            model2, _ = build_ref_model(M, align_mask_width=width)
            net = model2.denoising_net
        else:
            net = model.denoising_net
        y = net(t(x["motion"]), t(x["audio_feat"]), person, t(x["style"])[:, None], t(x["prev_motion"]),
                t(x["prev_audio"]), step, t(x["indicator"]))
        out[f"target_w{width}"] = y.numpy()
    dyn, stat, al = model.denoising_net(t(x["motion"]), t(x["audio_feat"]), person, t(x["style"])[:, None],
                                        t(x["prev_motion"]), t(x["prev_audio"]), step, t(x["indicator"]),
                                        keep_separate=True)
    out["dynamic"], out["static"], out["alphas"] = dyn.numpy(), stat.numpy()[:, :2], al.numpy()
    out["step"] = step.numpy()
    save("g3_denoiser", **out)


def g3_denoiser_options(M, SE, mc):
    """Two non-default switches of the denoiser the reference still reads: regularize_alpha='sigmoid'
    (model.py:13-17, 973-974) and no_use_learnable_pe (sinusoidal PE module, model.py:862-866, 950-953)."""
    B = 2
    out = {}
    step = torch.tensor([7, 433])
    for name, kw in (("sigmoid", dict(regularize_alpha="sigmoid")), ("sinpe", dict(no_use_learnable_pe=True))):
        model, args = build_ref_model(M, **kw)
        x = denoiser_inputs(B, args)
        person = torch.cat([t(x["shape"])[:, None], t(x["style"])[:, None]], dim=-1)
        net = model.denoising_net
        a = (t(x["motion"]), t(x["audio_feat"]), person, t(x["style"])[:, None], t(x["prev_motion"]),
             t(x["prev_audio"]), step, t(x["indicator"]))
        out[f"target_{name}"] = net(*a).numpy()
        dyn, stat, al = net(*a, keep_separate=True)
        out[f"alphas_{name}"] = al.numpy()
    out["step"] = step.numpy()
    save("g3_denoiser_options", **out)


def g3_forward(M, SE, mc):
    model, args = build_ref_model(M)
    B = 2
    x = denoiser_inputs(B, args, tag="fw")
    audio = synth.audio_clips(B, 64000, tag="fw_audio")
    out = {}
    # (a) raw audio, start tokens, no CFG masking
    torch.manual_seed(11)
    eps, target, _, afeat = model(t(x["motion"]), t(audio), t(x["shape"]), t(x["style"]), time_step=[3, 499],
                                  indicator=t(x["indicator"]), train_with_CFG=False)
    out.update(a_eps=eps.numpy(), a_target=target.numpy(), a_audio_feat=afeat.numpy()[:, ::2, ::3])
    # (b) feature input + prev frames + CFG masking on (replay the single rand draw)
    torch.manual_seed(12)
    flag = torch.rand(B)
    torch.manual_seed(12)
    eps, target, _, _ = model(t(x["motion"]), t(x["audio_feat"]), t(x["shape"]), t(x["style"]),
                              t(x["prev_motion"]), t(x["prev_audio"]), time_step=[250, 1],
                              indicator=t(x["indicator"]), train_with_CFG=True)
    out.update(b_eps=eps.numpy(), b_target=target.numpy(), b_flag=flag.numpy())
    save("g3_forward", **out)


def g3_style(M, SE, mc):
    args = ref_args()
    enc = SE.get_style_encoder(args, "vae2").eval()
    synth.load_synthetic(enc)
    out = {}
    for B, T in ((2, 100), (1, 60)):
        m = synth.motion_clips(B, T, tag="style_in")
        torch.manual_seed(5)
        z, mu, logvar = enc(t(m))
        torch.manual_seed(5)
        eps = torch.randn_like(mu)
        out[f"mu_{B}_{T}"], out[f"logvar_{B}_{T}"] = mu.numpy(), logvar.numpy()
        out[f"z_{B}_{T}"], out[f"eps_{B}_{T}"] = z.numpy(), eps.numpy()
    save("g3_style", **out)


def g3_sample(M, SE, mc):
    model, args = build_ref_model(M)
    B = 2
    x = denoiser_inputs(B, args, tag="sm")
    out = {}
    T = 3
    model.diffusion_sched = M.DiffusionSchedule(T, "cosine")
    xT = synth.normalish("sm/xT", (B, 100, 67))
    cases = {
        "inc": dict(cfg_mode="incremental", cfg_scale=1.15),
        "ind": dict(cfg_mode="independent", cfg_scale=[1.3, 0.9]),
        "audio_only": dict(cfg_cond=["audio"], cfg_scale=2.0),
        "nocfg": dict(cfg_cond=[]),
        "dt": dict(cfg_mode="incremental", cfg_scale=1.4, dynamic_threshold=(0.9, 0.5, 2.0)),
        "flex": dict(cfg_mode="incremental", cfg_scale=1.15, flexibility=0.5),
    }
    for seed, (name, kw) in enumerate(cases.items()):
        torch.manual_seed(100 + seed)
        zs = [torch.randn(B, 100, 67) for _ in range(T - 1)]  # draws for t = T..2, in order
        torch.manual_seed(100 + seed)
        y, _, _ = model.sample(t(x["audio_feat"]), t(x["shape"]), t(x["style"]), t(x["prev_motion"]),
                               t(x["prev_audio"]), motion_at_T=t(xT), indicator=t(x["indicator"]), **kw)
        out[f"{name}_x0"] = y.numpy()
        out[f"{name}_z"] = np.stack([z.numpy() for z in zs])  # index 0 -> t=T
    # target='noise' variant
    model_n, _ = build_ref_model(M, target="noise")
    model_n.diffusion_sched = M.DiffusionSchedule(T, "linear")
    torch.manual_seed(321)
    zs = [torch.randn(B, 100, 67) for _ in range(T - 1)]
    torch.manual_seed(321)
    y, _, _ = model_n.sample(t(x["audio_feat"]), t(x["shape"]), t(x["style"]), motion_at_T=t(xT), cfg_scale=1.15,
                             indicator=t(x["indicator"]))
    out["noise_x0"], out["noise_z"] = y.numpy(), np.stack([z.numpy() for z in zs])
    # sample_separate (model.py:442-651).  The reference only works at batch size 1 here: its static-feature tiling
    # (model.py:983-984) multiplies the batch instead of matching it, so B > 1 raises a shape error.
    torch.manual_seed(555)
    zs = [torch.randn(1, 100, 67) for _ in range(T - 1)]
    torch.manual_seed(555)
    r = model.sample_separate(t(x["audio_feat"][:1]), t(x["shape"][:1]), t(x["style"][:1]), t(x["prev_motion"][:1]),
                              t(x["prev_audio"][:1]), motion_at_T=t(xT[:1]), indicator=t(x["indicator"][:1]),
                              cfg_scale=1.3)
    out["sep_z"] = np.stack([z.numpy() for z in zs])
    out["sep_x0"], out["sep_dyn"], out["sep_static"], out["sep_alpha"] = (r[0].numpy(), r[3].numpy(), r[4].numpy(),
                                                                            r[5].numpy())
    save("g3_sample", **out)


def _load_infer_coeffs():
    """Build the reference's infer_coeffs function object from its file's AST (this container only)."""
    src = open(os.path.join(REF, "inference.py")).read()
    tree = ast.parse(src)
    fn = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == "infer_coeffs"][0]
    mod = ast.Module(body=[fn], type_ignores=[])
    import torch.nn.functional as F
    ns = dict(torch=torch, F=F, math=math)
    exec(compile(mod, "<reference inference.py:infer_coeffs>", "exec"), ns)
    return ns["infer_coeffs"]


def g3_infer(M, SE, mc):
    infer_coeffs = _load_infer_coeffs()
    model, args = build_ref_model(M)
    T = 2
    model.diffusion_sched = M.DiffusionSchedule(T, "cosine")
    out = {}
    for S in (100000, 32000):
        audio = synth.audio_clips(1, S, tag="infer")[0]
        style = synth.normalish("infer/style", (1, args.d_style))
        shape = np.zeros((1, 1, 100), np.float32)
        # Record the reference's own normal draws in call order (window 0: x_T then one z per step
        # t>1; window i>0: z per step only).  HF's encoder also consumes torch.rand([]) per layer
        # (LayerDrop coin, drawn even in eval mode), so replay-by-reseeding would be fragile.
        draws = []
        orig_randn, orig_randn_like = torch.randn, torch.randn_like

        def rec_randn(*a, **k):
            r = orig_randn(*a, **k)
            draws.append(r.clone())
            return r

        def rec_randn_like(*a, **k):
            r = orig_randn_like(*a, **k)
            draws.append(r.clone())
            return r
        torch.manual_seed(77)
        torch.randn, torch.randn_like = rec_randn, rec_randn_like
        try:
            y = infer_coeffs(model, args, t(audio), t(shape), 640.0, t(style), cfg_scale=1.4, dynamic_threshold=None)
        finally:
            torch.randn, torch.randn_like = orig_randn, orig_randn_like
        out[f"coef_{S}"] = y.numpy()
        out[f"draws_{S}"] = np.stack([d.numpy() for d in draws])
    save("g3_infer", **out)


# --------------------------------------------------------------------------- G3 style-clip ingestion
INGEST_CASES = ((90, 30, "tensor"), (131, 30, "numpy"), (77, 24, "tensor"), (50, 25, "numpy"), (400, 30, "tensor"))


def ingest_inputs(n, tag="ingest"):
    """Synthetic style clip + corpus statistics (shared with tests/test_host_cpu.py / test_model_gpu.py)."""
    e = (synth.normalish(f"{tag}/{n}/exp", (n, 50)) * 1.5 + 0.2).astype(np.float32)
    h = (synth.normalish(f"{tag}/{n}/head", (n, 3)) * 0.3).astype(np.float32)
    st = {"exp_mean": synth.normalish(f"{tag}/exp_mean", (50,)) * 0.1, "exp_std": np.abs(synth.normalish(f"{tag}/exp_std", (50,))) + 0.5,
          "pose_mean": synth.normalish(f"{tag}/pose_mean", (3,)) * 0.1, "pose_std": np.abs(synth.normalish(f"{tag}/pose_std", (3,))) + 0.5}
    return e, h, {k: v.astype(np.float32) for k, v in st.items()}


def g3_ingest(M, SE, mc):
    """reference inference.py:109-183 `query_for_motion_coeff` itself (function object built from the file's AST: the script
    cannot be imported here), fed pickle files as its caller does: corpus statistics, expression code (a tensor) and head
    rotation (tensor or array), original fps 30 / 24 / 25 -> 25."""
    import pickle as pkl
    from scipy.interpolate import interp1d
    tree = ast.parse(open(os.path.join(REF, "inference.py")).read())
    fn = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == "query_for_motion_coeff"][0]
    ns = dict(torch=torch, np=np, pkl=pkl, interp1d=interp1d, argparse=argparse)
    exec(compile(ast.Module(body=[fn], type_ignores=[]), "<reference inference.py:query_for_motion_coeff>", "exec"), ns)
    out = {"cases": np.array([f"{n},{fps},{kind}" for n, fps, kind in INGEST_CASES])}
    with tempfile.TemporaryDirectory() as d:
        for n, fps, kind in INGEST_CASES:
            e, h, st = ingest_inputs(n)
            paths = {k: os.path.join(d, f"{k}_{n}.pkl") for k in ("stats", "exp", "head")}
            for k, obj in (("stats", {a: t(b) for a, b in st.items()}), ("exp", t(e)), ("head", t(h) if kind == "tensor" else h)):
                with open(paths[k], "wb") as f:
                    pkl.dump(obj, f)
            args = argparse.Namespace(coef_dict_path=paths["stats"])
            motion, shape = ns["query_for_motion_coeff"](args, paths["exp"], paths["head"], device="cpu", original_fps=fps, target_fps=25)
            out[f"motion_{n}_{fps}"] = motion.numpy()
            out[f"shape_{n}_{fps}"] = shape.numpy()
    save("g3_ingest", **out)


# --------------------------------------------------------------------------- G4 FLAME / rotations
def write_flame_asset(tmpdir):
    a = synth.flame_asset()
    pkl = os.path.join(tmpdir, "generic_model.pkl")
    with open(pkl, "wb") as f:
        pickle.dump({k: a[k] for k in ("f", "v_template", "shapedirs", "posedirs", "J_regressor",
                                       "kintree_table", "weights")}, f)
    lmk = dict(a["lmk"])
    lmk["dynamic_lmk_faces_idx"] = torch.from_numpy(lmk["dynamic_lmk_faces_idx"])
    lmk["dynamic_lmk_bary_coords"] = torch.from_numpy(lmk["dynamic_lmk_bary_coords"])
    npy = os.path.join(tmpdir, "landmark_embedding.npy")
    np.save(npy, lmk, allow_pickle=True)
    return pkl, npy


def flame_inputs(B, tag="flame"):
    return dict(shape=(0.5 * synth.normalish(f"{tag}/shape", (B, 100))).astype(np.float32),
                exp=(0.5 * synth.normalish(f"{tag}/exp", (B, 50))).astype(np.float32),
                pose=(0.4 * synth.normalish(f"{tag}/pose", (B, 6))).astype(np.float32))


def g4_flame(M, SE, mc):
    from utils import flame as FL, lbs as LBS
    with tempfile.TemporaryDirectory() as td:
        pkl, npy = write_flame_asset(td)
        cfg = FL.FLAMEConfig
        cfg.flame_model_path, cfg.flame_lmk_embedding_path = pkl, npy
        fl = FL.FLAME(cfg).eval()
    B = 8
    x = flame_inputs(B)
    x["pose"][0] = 0.0            # zero rotation (eps placement of batch_rodrigues)
    x["pose"][1, :3] = [0.0, 1.2, 0.0]   # large yaw -> dynamic landmark LUT > 39 clamp
    x["pose"][2, :3] = [0.0, -1.2, 0.0]  # negative yaw branch
    x["pose"][3, :3] = [0.0, -0.3, 0.0]
    v, lm2d, lm3d = fl(t(x["shape"]), t(x["exp"]), t(x["pose"]))
    v_nog, _, _ = fl(t(x["shape"]), t(x["exp"]), t(x["pose"]), ignore_global_rot=True, return_lm2d=False,
                     return_lm3d=False)
    vn = v.numpy()
    out = dict(pose=x["pose"], verts_sub=vn[:, ::79], verts_sum=vn.sum(axis=1, dtype=np.float64),
               verts_abs_sum=np.abs(vn).sum(axis=1, dtype=np.float64), lm2d=lm2d.numpy(), lm3d=lm3d.numpy(),
               verts_nog_sub=v_nog.numpy()[:, ::79])
    rv = np.concatenate([np.zeros((1, 3), np.float32), np.array([[math.pi, 0, 0], [0, 0, 1e-9], [1e-4, -2e-4, 3e-4]],
                        np.float32), (1.5 * synth.normalish("rodrigues", (12, 3))).astype(np.float32)])
    out["rodrigues_in"], out["rodrigues_out"] = rv, LBS.batch_rodrigues(t(rv)).numpy()
    save("g4_flame", **out)


def g4_lbs_blocks(M, SE, mc):
    """The building blocks of the reference's utils/lbs.py on the synthetic FLAME asset: blend_shapes, vertices2joints,
    transform_mat, batch_rigid_transform, find_dynamic_lmk_idx_and_bcoords."""
    from utils import flame as FL, lbs as LBS
    with tempfile.TemporaryDirectory() as td:
        pkl, npy = write_flame_asset(td)
        cfg = FL.FLAMEConfig
        cfg.flame_model_path, cfg.flame_lmk_embedding_path = pkl, npy
        fl = FL.FLAME(cfg).eval()
    B = 4
    x = flame_inputs(B, tag="blocks")
    x["pose"][1, :3] = [0.0, 1.2, 0.0]
    x["pose"][2, :3] = [0.0, -0.4, 0.0]
    betas = torch.cat([t(x["shape"]), t(x["exp"])], dim=1)
    out = {}
    bs = LBS.blend_shapes(betas, fl.shapedirs)
    out["blend_sub"] = bs.numpy()[:, ::79]
    v_shaped = fl.v_template.unsqueeze(0) + bs
    J = LBS.vertices2joints(fl.J_regressor, v_shaped)
    out["joints"] = J.numpy()
    full_pose = torch.cat([t(x["pose"])[:, :3], fl.neck_pose.expand(B, -1), t(x["pose"])[:, 3:], fl.eye_pose.expand(B, -1)], dim=1)
    rot = LBS.batch_rodrigues(full_pose.view(-1, 3)).view(B, -1, 3, 3)
    posed, rel = LBS.batch_rigid_transform(rot, J, fl.parents)
    out["full_pose"], out["posed"], out["rel"] = full_pose.numpy(), posed.numpy(), rel.numpy()
    out["tmat"] = LBS.transform_mat(rot[:, 1], J[:, 1].unsqueeze(-1)).numpy()
    # the module-level function only runs at batch size 1 (its bmm against a (1, 3, 3) identity, utils/lbs.py:83)
    pairs = [LBS.find_dynamic_lmk_idx_and_bcoords(v_shaped[b:b + 1], full_pose[b:b + 1], fl.dynamic_lmk_faces_idx,
                                                  fl.dynamic_lmk_bary_coords, fl.neck_kin_chain) for b in range(B)]
    out["dyn_idx"] = torch.cat([p[0] for p in pairs]).numpy()
    out["dyn_bary"] = torch.cat([p[1] for p in pairs]).numpy()
    save("g4_lbs_blocks", **out)


def g4_rotations(M, SE, mc):
    from utils import rotation_conversions as RC
    n = 32
    aa = (1.2 * synth.normalish("rot/aa", (n, 3))).astype(np.float32)
    aa[0] = 0.0
    aa[1] = [1e-7, -2e-7, 1e-7]       # small-angle Taylor branch
    aa[2] = [math.pi, 0.0, 0.0]
    q = synth.normalish("rot/q", (n, 4))
    q2 = synth.normalish("rot/q2", (n, 4))
    pts = synth.normalish("rot/pts", (n, 3))
    d6 = synth.normalish("rot/d6", (n, 6))
    eul = (1.0 * synth.normalish("rot/eul", (n, 3))).astype(np.float32)
    out = dict(aa=aa, q=q, q2=q2, pts=pts, d6=d6, eul=eul)
    R = RC.axis_angle_to_matrix(t(aa))
    out["axis_angle_to_matrix"] = R.numpy()
    out["axis_angle_to_quaternion"] = RC.axis_angle_to_quaternion(t(aa)).numpy()
    out["quaternion_to_matrix"] = RC.quaternion_to_matrix(t(q)).numpy()
    out["matrix_to_quaternion"] = RC.matrix_to_quaternion(R).numpy()
    out["quaternion_to_axis_angle"] = RC.quaternion_to_axis_angle(t(q)).numpy()
    out["matrix_to_axis_angle"] = RC.matrix_to_axis_angle(R).numpy()
    out["rotation_6d_to_matrix"] = RC.rotation_6d_to_matrix(t(d6)).numpy()
    out["matrix_to_rotation_6d"] = RC.matrix_to_rotation_6d(R).numpy()
    out["axis_angle_to_rotation_6d"] = RC.axis_angle_to_rotation_6d(t(aa)).numpy()
    out["quaternion_raw_multiply"] = RC.quaternion_raw_multiply(t(q), t(q2)).numpy()
    out["quaternion_multiply"] = RC.quaternion_multiply(t(q), t(q2)).numpy()
    out["quaternion_invert"] = RC.quaternion_invert(t(q)).numpy()
    out["quaternion_apply"] = RC.quaternion_apply(t(q), t(pts)).numpy()
    out["standardize_quaternion"] = RC.standardize_quaternion(t(q)).numpy()
    for conv in ("XYZ", "ZYX", "YXZ", "XYX", "ZXZ"):
        Re = RC.euler_angles_to_matrix(t(eul), conv)
        out[f"euler_angles_to_matrix_{conv}"] = Re.numpy()
        out[f"matrix_to_euler_angles_{conv}"] = RC.matrix_to_euler_angles(Re, conv).numpy()
    save("g4_rotations", **out)


def g5_losses_no_constrain_prev(M, SE, mc):
    """The reference's (deprecated but live) --no_constrain_prev mode of both loss functions (utils/common.py:245,
    382, 481, 561): the previous-window part of the prediction is replaced by ground truth and masked out."""
    from utils import common as C
    from utils import flame as FL
    args = ref_args(no_constrain_prev=True)
    N = 3
    gt = synth.normalish("loss/gt", (N, 100, 67))
    prev = synth.normalish("loss/prev", (N, 10, 67))
    target = synth.normalish("loss/target", (N, 110, 67))
    end_idx = torch.tensor([100, 37, 1])
    out = {}
    for start in (True, False):
        for use_end in (False, True):
            r = C.compute_loss_no_vert(args, start, None, t(gt), None, t(target), t(prev), None, None,
                                       end_idx=end_idx if use_end else None)
            out[f"nv_{int(start)}_{int(use_end)}"] = np.array([np.nan if v is None else float(v) for v in r], np.float64)
    with tempfile.TemporaryDirectory() as td:
        pkl, npy = write_flame_asset(td)
        cfg = FL.FLAMEConfig
        cfg.flame_model_path, cfg.flame_lmk_embedding_path = pkl, npy
        fl = FL.FLAME(cfg).eval()
    L = 12
    argsv = ref_args(n_motions=L, n_prev_motions=4, no_constrain_prev=True)
    gt54 = (0.5 * synth.normalish("loss/gt54", (2, L, 54))).astype(np.float32)
    prev54 = (0.5 * synth.normalish("loss/prev54", (2, 4, 54))).astype(np.float32)
    tgt54 = (0.5 * synth.normalish("loss/tgt54", (2, L + 4, 54))).astype(np.float32)
    shape = (0.5 * synth.normalish("loss/shape", (2, 100))).astype(np.float32)
    stats = {"exp_mean": t(0.1 * synth.normalish("st/em", (50,))), "exp_std": t(1 + 0.1 * np.abs(synth.normalish("st/es", (50,)))),
             "pose_mean": t(0.05 * synth.normalish("st/pm", (6,))), "pose_std": t(1 + 0.1 * np.abs(synth.normalish("st/ps", (6,)))),
             "shape_mean": t(np.zeros(100, np.float32)), "shape_std": t(np.ones(100, np.float32))}
    for start in (True, False):
        r = C.compute_loss(argsv, start, t(shape), t(gt54), None, t(tgt54), t(prev54), stats, fl,
                           end_idx=torch.tensor([L, 5]))
        out[f"vert_{int(start)}"] = np.array([np.nan if v is None else float(v) for v in r], np.float64)
    save("g5_losses_no_constrain_prev", **out)


# --------------------------------------------------------------------------- G5 losses / scheduler
def g5_losses(M, SE, mc):
    from utils import common as C
    from utils.scheduler import GradualWarmupScheduler
    args = ref_args()
    N = 3
    gt = synth.normalish("loss/gt", (N, 100, 67))
    prev = synth.normalish("loss/prev", (N, 10, 67))
    target = synth.normalish("loss/target", (N, 110, 67))
    end_idx = torch.tensor([100, 37, 1])
    out = {}
    for start in (True, False):
        for use_end in (False, True):
            r = C.compute_loss_no_vert(args, start, None, t(gt), None, t(target), t(prev), None, None,
                                       end_idx=end_idx if use_end else None)
            out[f"nv_{int(start)}_{int(use_end)}"] = np.array([np.nan if v is None else float(v) for v in r], np.float64)
    args1 = ref_args(criterion="l1")
    r = C.compute_loss_no_vert(args1, False, None, t(gt), None, t(target), t(prev), None, None, end_idx=end_idx)
    out["nv_l1"] = np.array([np.nan if v is None else float(v) for v in r], np.float64)
    mu = synth.normalish("loss/mu", (N, 256))
    logvar = (0.3 * synth.normalish("loss/logvar", (N, 256))).astype(np.float32)
    out["kl"] = np.float64(C.compute_KL_loss(t(mu), t(logvar)))
    # vertex-space variant (legacy 54-d motion) through the reference FLAME on the synthetic asset
    from utils import flame as FL
    with tempfile.TemporaryDirectory() as td:
        pkl, npy = write_flame_asset(td)
        cfg = FL.FLAMEConfig
        cfg.flame_model_path, cfg.flame_lmk_embedding_path = pkl, npy
        fl = FL.FLAME(cfg).eval()
    L = 12
    argsv = ref_args(n_motions=L, n_prev_motions=4)
    gt54 = (0.5 * synth.normalish("loss/gt54", (2, L, 54))).astype(np.float32)
    prev54 = (0.5 * synth.normalish("loss/prev54", (2, 4, 54))).astype(np.float32)
    tgt54 = (0.5 * synth.normalish("loss/tgt54", (2, L + 4, 54))).astype(np.float32)
    shape = (0.5 * synth.normalish("loss/shape", (2, 100))).astype(np.float32)
    stats = {"exp_mean": t(0.1 * synth.normalish("st/em", (50,))), "exp_std": t(1 + 0.1 * np.abs(synth.normalish("st/es", (50,)))),
             "pose_mean": t(0.05 * synth.normalish("st/pm", (6,))), "pose_std": t(1 + 0.1 * np.abs(synth.normalish("st/ps", (6,)))),
             "shape_mean": t(np.zeros(100, np.float32)), "shape_std": t(np.ones(100, np.float32))}
    for start in (True, False):
        r = C.compute_loss(argsv, start, t(shape), t(gt54), None, t(tgt54), t(prev54), stats, fl,
                           end_idx=torch.tensor([L, 5]))
        out[f"vert_{int(start)}"] = np.array([np.nan if v is None else float(v) for v in r], np.float64)
    # coefficient glue
    cd = C.get_coef_dict(t(gt54), t(shape), stats, with_global_pose=False)
    out["coef_exp"], out["coef_pose"], out["coef_shape"] = cd["exp"].numpy(), cd["pose"].numpy(), cd["shape"].numpy()
    verts = C.coef_dict_to_vertices(cd, fl, flame_batch_size=7)
    out["coef_verts_sub"] = verts.numpy()[:, :, ::79]
    out["motion_coef"] = C.get_motion_coef({"exp": cd["exp"], "pose": cd["pose"]}, "aa", with_global_pose=False).numpy()
    # truncation with a fixed end index (the random draw is host RNG)
    a = synth.audio_clips(2, 64000, tag="trunc")
    mc_ = synth.motion_clips(2, tag="trunc_m")
    e = torch.tensor([3, 77])
    out["trunc_audio_zero"] = C._truncate_audio(t(a), (e * 640).long(), "zero").numpy()[:, ::97]
    out["trunc_audio_rep"] = C._truncate_audio(t(a), (e * 640).long(), "replicate").numpy()[:, ::97]
    cdt = C._truncate_coef_dict({"exp": t(mc_[..., :50]), "pose_any": t(mc_[..., 50:])}, e, "replicate")
    out["trunc_motion_rep"] = torch.cat([cdt["exp"], cdt["pose_any"]], -1).numpy()
    # scheduler trace
    p = torch.nn.Parameter(torch.zeros(1))
    opt = torch.optim.Adam([{"params": [p], "lr": 2e-5}])
    sch = GradualWarmupScheduler(opt, 1, 10)
    lrs = []
    for _ in range(15):
        opt.step(); sch.step(); lrs.append(opt.param_groups[0]["lr"])
    out["warmup_lrs"] = np.array(lrs, np.float64)
    # Adam reference trace (torch.optim.Adam, lr 2e-5) on a fixed gradient sequence
    w = torch.nn.Parameter(t(synth.normalish("adam/w", (1000,))).clone())
    opt = torch.optim.Adam([w], lr=2e-3)
    for i in range(3):
        w.grad = t(synth.normalish(f"adam/g{i}", (1000,)))
        opt.step()
    out["adam_w3"] = w.detach().numpy()
    save("g5_losses", **out)


# --------------------------------------------------------------------------- G6 one fixed-noise training step
G6_PARAMS = [
    "denoising_net.PE", "denoising_net.person_proj.weight", "denoising_net.feature_proj.weight",
    "denoising_net.diff_step_map.0.weight", "denoising_net.transformer.layers.0.self_attn.in_proj_weight",
    "denoising_net.transformer.layers.0.multihead_attn.in_proj_weight",
    "denoising_net.transformer.layers.1.linear2.weight", "denoising_net.transformer.layers.1.norm3.weight",
    "denoising_net.motion_dec.2.bias", "denoising_net.static_feature_mapping.3.2.weight",
    "audio_feature_map.weight", "start_motion_feat", "start_audio_feat",
    "audio_encoder.encoder.layers.1.feed_forward.output_dense.weight",
    "audio_encoder.encoder.layers.0.attention.q_proj.weight", "audio_encoder.encoder.layers.0.layer_norm.bias",
    "audio_encoder.encoder.layer_norm.weight", "audio_encoder.encoder.pos_conv_embed.conv.bias",
    "audio_encoder.encoder.pos_conv_embed.conv.weight_g", "audio_encoder.encoder.pos_conv_embed.conv.weight_v",
    "audio_encoder.feature_projection.projection.weight", "audio_encoder.feature_projection.layer_norm.weight",
]
G6_STYLE = ["input_layers.1.weight", "input_layers.5.bias", "encoder.self_attn.in_proj_weight", "encoder.linear1.weight",
            "output_layers.7.bias"]


G6_FULL_EXTRA = [
    "denoising_net.transformer.layers.7.linear1.weight", "denoising_net.transformer.layers.4.multihead_attn.out_proj.weight",
    "denoising_net.transformer.layers.3.norm2.bias", "audio_encoder.encoder.layers.11.feed_forward.intermediate_dense.weight",
    "audio_encoder.encoder.layers.6.attention.v_proj.weight", "audio_encoder.encoder.layers.3.final_layer_norm.weight",
]


def g6_train_full(M, SE, mc):
    """g6_train at FULL depth (12 encoder + 8 decoder layers): for every listed parameter the gradient norm and 64
    entries sampled across the tensor (element-wise parity, not only norms)."""
    return g6_train(M, SE, mc, enc_layers=12, dec_layers=8, name="g6_train_full", params=G6_PARAMS + G6_FULL_EXTRA,
                    sample=64)


def g6_train(M, SE, mc, enc_layers=2, dec_layers=2, name="g6_train", params=None, sample=0):
    """Window-0 training forward of the reference in eval mode (no dropout) WITH gradients: fixed t / eps, no CFG
    masking, style from the VAE mean path; loss = weighted parameter-space terms + KL; records losses and the
    gradients' norms and leading entries for parameters spread over every component."""
    from transformers import Wav2Vec2Config
    import utils.wav2vec2 as w2
    from utils import common as C
    params = params or G6_PARAMS
    w2.Wav2Vec2Model.from_pretrained = classmethod(
        lambda cls, name, **kw: cls(Wav2Vec2Config(attn_implementation="eager", num_hidden_layers=enc_layers)))
    torch.set_grad_enabled(True)
    try:
        args = ref_args(n_layers=dec_layers)
        model = M.get_diffusion_model(args, device="cpu").eval()
        synth.load_synthetic(model)
        se = SE.get_style_encoder(args, "vae2").eval()
        synth.load_synthetic(se)
        B = 2
        audio = t(synth.audio_clips(B, 64000, tag="g6_audio"))
        motion = t(synth.motion_clips(B, tag="g6_motion"))
        eps = synth.normalish("g6/eps", (B, 100, 67))
        zst = synth.normalish("g6/zstyle", (B, 256))
        shape = torch.zeros(B, 100)
        ind = torch.ones(B, 100)
        ind[1, 70:] = 0
        end_idx = torch.tensor([100, 70])
        with mock_randn_like([t(zst), t(eps)]):
            style, mu, logvar = se(motion)
            noise, target, _, _ = model(motion, audio, shape, style, time_step=[7, 311], indicator=ind,
                                        train_with_CFG=False)
        losses = C.compute_loss_no_vert(args, True, None, motion, noise, target, None, None, None, end_idx=end_idx)
        kl = C.compute_KL_loss(mu, logvar)
        wts = [args.l_vert, args.l_vel * 4.5e-8 * 1e8, args.l_smooth * 4e-7 * 1e7, args.l_head_angle, args.l_head_vel,
               args.l_head_smooth]
        total = sum(w * l for w, l in zip(wts, losses[:6])) + 1e-3 * kl
        total.backward()
        out = dict(losses=np.array([float(l) for l in losses[:6]] + [float(kl), float(total)], np.float64),
                   target=target.detach().numpy(), mu=mu.detach().numpy())
        named = dict(model.named_parameters())
        def pick(gr):
            flat = gr.reshape(-1)
            if not sample:
                return flat[:8].numpy().astype(np.float64)
            return flat[::max(1, flat.numel() // sample) | 1][:sample].numpy().astype(np.float64)   # odd stride: walks every axis
        for k in params:
            kk = k.replace("weight_g", "parametrizations.weight.original0").replace("weight_v", "parametrizations.weight.original1")
            gr = named[kk].grad
            out["gn/" + k] = np.float64(gr.double().norm())
            out["g8/" + k] = pick(gr)
        snamed = dict(se.named_parameters())
        for k in G6_STYLE:
            gr = snamed[k].grad
            out["sn/" + k] = np.float64(gr.double().norm())
            out["s8/" + k] = pick(gr)
        save(name, **out)
    finally:
        torch.set_grad_enabled(False)
        w2.Wav2Vec2Model.from_pretrained = classmethod(
            lambda cls, name, **kw: cls(Wav2Vec2Config(attn_implementation="eager")))


def g8_vertex_grad(M, SE, mc):
    """Gradient of the reference's vertex-space training loss (training_script.py:167-176 -> utils/common.py:456-620
    -> utils/flame.py / utils/lbs.py, differentiated by the reference's own autograd) with respect to the predicted
    motion coefficients, on the synthetic FLAME asset: both windows, truncated end indices, all loss weights > 0."""
    from utils import common as C
    from utils import flame as FL
    with tempfile.TemporaryDirectory() as td:
        pkl, npy = write_flame_asset(td)
        cfg = FL.FLAMEConfig
        cfg.flame_model_path, cfg.flame_lmk_embedding_path = pkl, npy
        fl = FL.FLAME(cfg).eval()
    L, P, N = 12, 4, 3
    args = ref_args(n_motions=L, n_prev_motions=P, use_vertex_space=True)
    gt = (0.5 * synth.normalish("vgrad/gt", (N, L, 54))).astype(np.float32)
    prev = (0.5 * synth.normalish("vgrad/prev", (N, P, 54))).astype(np.float32)
    tgt = (0.5 * synth.normalish("vgrad/tgt", (N, L + P, 54))).astype(np.float32)
    shape = (0.5 * synth.normalish("vgrad/shape", (N, 100))).astype(np.float32)
    stats = {"exp_mean": t(0.1 * synth.normalish("st/em", (50,))), "exp_std": t(1 + 0.1 * np.abs(synth.normalish("st/es", (50,)))),
             "pose_mean": t(0.05 * synth.normalish("st/pm", (6,))), "pose_std": t(1 + 0.1 * np.abs(synth.normalish("st/ps", (6,)))),
             "shape_mean": t(np.zeros(100, np.float32)), "shape_std": t(np.ones(100, np.float32))}
    end_idx = torch.tensor([L, 5, 9])
    wts = dict(noise=1.0, vert=args.l_vert, vel=args.l_vel, smooth=args.l_smooth, head_angle=args.l_head_angle,
               head_vel=args.l_head_vel, head_smooth=args.l_head_smooth, head_trans=args.l_head_trans)
    out = {"weights": np.array([wts[k] for k in ("noise", "vert", "vel", "smooth", "head_angle", "head_vel", "head_smooth",
                                                   "head_trans")], np.float64)}
    torch.set_grad_enabled(True)
    try:
        for start in (True, False):
            for use_end in (False, True):
                tg = t(tgt).clone().requires_grad_(True)
                ld = C.compute_loss(args, start, t(shape), t(gt), None, tg, t(prev), stats, fl,
                                    end_idx=end_idx if use_end else None, return_dict=True)
                total = sum(wts[k] * v for k, v in ld.items() if v is not None and not isinstance(v, int))
                total.backward()
                key = f"{int(start)}_{int(use_end)}"
                out["loss_" + key] = np.array([np.nan if (v is None) else float(v) for v in
                                               (ld[k] for k in ("noise", "vert", "vel", "smooth", "head_angle", "head_vel",
                                                                "head_smooth", "head_trans"))], np.float64)
                out["grad_" + key] = tg.grad.numpy().copy()
        # the FLAME pass alone: d(sum of verts * probe) / d(exp, pose)
        B = 6
        x = flame_inputs(B, tag="vgrad_flame")
        ex, po = t(x["exp"]).clone().requires_grad_(True), t(x["pose"]).clone().requires_grad_(True)
        probe = synth.normalish("vgrad/probe", (B, 5023, 3))
        v, _, _ = fl(t(x["shape"]), ex, po, return_lm2d=False, return_lm3d=False)
        (v * t(probe)).sum().backward()
        out["flame_dexp"], out["flame_dpose"] = ex.grad.numpy().copy(), po.grad.numpy().copy()
    finally:
        torch.set_grad_enabled(False)
    save("g8_vertex_grad", **out)


class mock_randn_like:
    """Feed a fixed sequence of tensors to torch.randn_like (style VAE eps, then diffusion eps)."""

    def __init__(self, seq):
        self.seq = list(seq)

    def __enter__(self):
        self.orig = torch.randn_like
        torch.randn_like = lambda x, *a, **k: self.seq.pop(0).to(x.dtype)
        return self

    def __exit__(self, *a):
        torch.randn_like = self.orig


def g1_specaug(M, SE, mc):
    """SpecAugment time-mask indices: the reference wav2vec2 wrapper's own routine (utils/wav2vec2.py:17-53) and the
    installed transformers' `_compute_mask_indices` (what HubertModel._mask_hidden_states calls, utils/hubert.py:35),
    both seeded through the global numpy RNG they consume."""
    import utils.wav2vec2 as w2
    from transformers.models.wav2vec2.modeling_wav2vec2 import _compute_mask_indices as hf_mask
    out = {"transformers_version": np.array(transformers.__version__)}
    cases = [(4, 200, 0.05, 10, 2), (32, 200, 0.05, 10, 2), (2, 500, 0.05, 10, 2), (3, 100, 0.3, 10, 2), (2, 40, 0.65, 10, 0)]
    out["cases"] = np.array(cases, dtype=np.float64)
    for i, (b, T, prob, length, mn) in enumerate(cases):
        for seed in (0, 1234):
            np.random.seed(seed)
            out[f"ref_{i}_{seed}"] = w2._compute_mask_indices((b, T), prob, length, min_masks=mn)
            np.random.seed(seed)
            out[f"hf_{i}_{seed}"] = hf_mask((b, T), prob, length, min_masks=mn)
    save("g1_specaug", **out)


# --------------------------------------------------------------------------- G9 public signatures of the drop-in surface
SIGNATURE_TARGETS = {
    # module (as imported by import_reference / by name) -> callables; "Class.method" walks one attribute
    "model": ["get_diffusion_model", "MSMD.__init__", "MSMD.forward", "MSMD.extract_audio_feature", "MSMD.extract_audio_768_feature",
              "MSMD.sample", "MSMD.sample_separate", "MSMD.sample_with_guide", "DenoisingNetwork_MSMD.__init__",
              "DenoisingNetwork_MSMD.forward", "DiffusionSchedule.__init__", "DiffusionSchedule.uniform_sample_t",
              "DiffusionSchedule.get_sigmas"],
    "style_encoder": ["get_style_encoder", "StyleEncoder_VAE2.__init__", "StyleEncoder_VAE2.forward", "StyleEncoder_VAE2.sample"],
    "utils.wav2vec2": ["Wav2Vec2Model.forward", "linear_interpolation", "_compute_mask_indices"],
    "utils.hubert": ["HubertModel.forward", "linear_interpolation"],
    "utils.model_common": ["pad_audio", "enc_dec_mask", "PositionalEncoding.__init__", "PositionalEncoding.forward"],
    "utils.flame": ["FLAME.__init__", "FLAME.forward", "FLAME.seletec_3d68", "FLAME._find_dynamic_lmk_idx_and_bcoords"],
    "utils.lbs": ["lbs", "blend_shapes", "vertices2joints", "batch_rodrigues", "transform_mat", "batch_rigid_transform",
                  "vertices2landmarks", "rot_mat_to_euler", "find_dynamic_lmk_idx_and_bcoords"],
    "utils.rotation_conversions": None,      # every public function
    "utils.common": ["get_pose_input", "get_motion_coef", "get_coef_dict", "coef_dict_to_vertices", "compute_loss_no_vert",
                     "compute_loss", "compute_KL_loss", "truncate_motion_coef_and_audio"],
    "utils.scheduler": ["GradualWarmupScheduler.__init__", "GradualWarmupScheduler.get_lr", "GradualWarmupScheduler.step"],
    "inference": ["infer_coeffs", "load_args", "load_model", "query_for_motion_coeff"],     # by AST: the script cannot be imported here
}


def signature_record(fn):
    import inspect
    out = []
    for p in inspect.signature(fn).parameters.values():
        out.append([p.name, p.kind.name, None if p.default is inspect.Parameter.empty else repr(p.default)])
    return out


def g9_signatures(M, SE, mc):
    """Parameter lists (name, kind, default) of every callable of the drop-in surface, read from the reference's own
    function objects; tests/test_host_cpu.py holds the product's callables against them."""
    import importlib
    import inspect
    import json
    rec = {}
    for modname, names in SIGNATURE_TARGETS.items():
        if modname == "inference":
            tree = ast.parse(open(os.path.join(REF, "inference.py")).read())
            import argparse as _ap
            import torch.nn.functional as F
            for fn in [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in names]:
                ns = dict(torch=torch, F=F, math=math, argparse=_ap, np=np)
                exec(compile(ast.Module(body=[fn], type_ignores=[]), "<reference inference.py>", "exec"), ns)
                rec[f"inference.{fn.name}"] = signature_record(ns[fn.name])
            continue
        mod = importlib.import_module(modname)
        if names is None:
            names = [n for n, o in vars(mod).items() if inspect.isfunction(o) and o.__module__ == mod.__name__ and not n.startswith("_")]
        for name in names:
            obj = mod
            for part in name.split("."):
                obj = getattr(obj, part)
            rec[f"{modname}.{name}"] = signature_record(obj)
    path = os.path.join(HERE, "g9_signatures.json")
    with open(path, "w") as f:
        json.dump({"_versions": VERSIONS, "signatures": rec}, f, indent=1, sort_keys=True)
    print(f"wrote {path}  ({len(rec)} callables)")


# --------------------------------------------------------------------------- G10 fresh-parameter initialisation
INIT_CASES = (("default", {}),
              ("variant", dict(no_use_learnable_pe=True, guiding_conditions="audio", num_of_basis=2, use_indicator=False,
                               n_layers=3, d_style=128, dataset_type="flame_mead_ravdess")))


def _tensor_record(v):
    import zlib
    a = np.ascontiguousarray(v.detach().float().numpy())
    d = a.astype(np.float64).ravel()
    return (np.array([d.mean(), d.std(), d.min(), d.max()]), np.uint32(zlib.crc32(a.tobytes())),
            a.ravel()[:8].copy())


def g10_init(M, SE, mc):
    """What the reference's CONSTRUCTORS leave in every parameter they create themselves (everything but the pretrained
    audio encoder): reference model.py:115-138, 856-908 and style_encoder.py:133-176, i.e. the defaults of the torch.nn
    layers they instantiate, drawn from the CPU generator.  The generator is seeded at the moment `from_pretrained` returns
    (model) / right before the constructor (style encoder), so the record is (a) per-tensor mean / std / min / max and
    (b) the exact draw (crc32 of the bytes + the first 8 values) a product constructor must reproduce from the same state."""
    import utils.wav2vec2 as w2
    from transformers import Wav2Vec2Config
    saved = w2.Wav2Vec2Model.__dict__["from_pretrained"]
    out = {}
    try:
        for case, kw in INIT_CASES:
            for seed in (0, 1):
                def seeded(cls, name, seed=seed, **_kw):
                    m = cls(Wav2Vec2Config(attn_implementation="eager", num_hidden_layers=1))
                    torch.manual_seed(seed)
                    return m
                w2.Wav2Vec2Model.from_pretrained = classmethod(seeded)
                args = ref_args(**kw)
                model = M.get_diffusion_model(args, device="cpu")
                torch.manual_seed(seed)
                se = SE.get_style_encoder(args, "vae2")
                for prefix, mod in (("model", model), ("style", se)):
                    keys = [k for k, _ in mod.named_parameters() if not k.startswith("audio_encoder.")]
                    out[f"{case}/{seed}/{prefix}/keys"] = np.array(keys)
                    recs = [_tensor_record(dict(mod.named_parameters())[k]) for k in keys]
                    out[f"{case}/{seed}/{prefix}/stats"] = np.stack([r[0] for r in recs])
                    out[f"{case}/{seed}/{prefix}/crc"] = np.array([r[1] for r in recs], dtype=np.uint32)
                    out[f"{case}/{seed}/{prefix}/head"] = np.stack([np.pad(r[2], (0, 8 - len(r[2]))) for r in recs])
                    out[f"{case}/{seed}/{prefix}/numel"] = np.array([dict(mod.named_parameters())[k].numel() for k in keys])
    finally:
        w2.Wav2Vec2Model.from_pretrained = saved
    import json
    out["cases"] = np.array(json.dumps(dict(INIT_CASES)))
    save("g10_init", **out)


ALL = dict(g10_init=g10_init, g3_ingest=g3_ingest, g9_signatures=g9_signatures, g1_specaug=g1_specaug, g7_dataset=g7_dataset, g2_lr_schedule=g2_lr_schedule, g3_audio_large=g3_audio_large, g1_index=g1_index, g2_schedule=g2_schedule, g3_audio=g3_audio, g3_denoiser=g3_denoiser,
           g3_forward=g3_forward, g3_style=g3_style, g3_sample=g3_sample, g3_infer=g3_infer,
           g4_flame=g4_flame, g4_rotations=g4_rotations, g5_losses=g5_losses, g6_train=g6_train,
           g5_losses_no_constrain_prev=g5_losses_no_constrain_prev, g3_denoiser_options=g3_denoiser_options, g4_lbs_blocks=g4_lbs_blocks,
           g8_vertex_grad=g8_vertex_grad, g6_train_full=g6_train_full)

if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("names", nargs="*", default=list(ALL))
    ns = ap.parse_args()
    torch.set_num_threads(8)
    mods = import_reference()
    for n in ns.names:
        print("==", n)
        ALL[n](*mods)
