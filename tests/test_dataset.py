"""Training-data tensor contract: oracle restatement against items / batches recorded from
the reference's own DatasetPickle (CPU), and the HBM-resident HIP loader against both (GPU)."""
import numpy as np
import pytest

from msmd_amd import synth
from oracle import dataset as ods

from conftest import load_golden


def raw_clips():
    """Same synthetic corpus as tests/golden/make_goldens.py::dataset_raw_clips."""
    raw = {}
    for name, n30 in (("long_a", 400), ("long_b", 301), ("exact", 252), ("short_a", 200), ("short_b", 131), ("tiny", 40)):
        n25 = int(round(n30 / 30 * 25))
        S = int(n25 * 640) + {"long_a": 37, "long_b": -211, "exact": 0, "short_a": 5, "short_b": -400, "tiny": 123}[name]
        raw[name] = {"audio": (0.3 * synth.normalish(f"ds/{name}/audio", (S,)) + 0.05).astype(np.float32),
                     "expression_code": synth.normalish(f"ds/{name}/exp", (n30, 64)).astype(np.float64) * 1.5 + 0.2,
                     "head_orientation": synth.normalish(f"ds/{name}/head", (n30, 3)).astype(np.float64) * 0.3}
    return raw


def coef_stats():
    st = {"exp_mean": synth.normalish("ds/exp_mean", (64,)) * 0.1, "exp_std": np.abs(synth.normalish("ds/exp_std", (64,))) + 0.5,
          "pose_mean": synth.normalish("ds/pose_mean", (3,)) * 0.1, "pose_std": np.abs(synth.normalish("ds/pose_std", (3,))) + 0.5}
    return {k: v.astype(np.float32) for k, v in st.items()}


def check_batch(g, mode, audio, motion, stats, tol):
    assert np.abs(audio[0][:, ::13] - g[f"{mode}_audio0"]).max() <= tol
    assert np.abs(audio[1][:, ::13] - g[f"{mode}_audio1"]).max() <= tol
    assert np.abs(audio[0][:, :64] - g[f"{mode}_audio0_head"]).max() <= tol
    assert np.abs(audio[1][:, -4000:] - g[f"{mode}_audio1_tail"]).max() <= tol
    assert np.abs(motion[0] - g[f"{mode}_motion0"]).max() <= tol
    assert np.abs(motion[1] - g[f"{mode}_motion1"]).max() <= tol
    assert np.abs(np.asarray(stats, np.float64) - g[f"{mode}_stats"]).max() < 1e-6


def test_oracle_items_and_collate_match_reference_dataset():
    g = load_golden("g7_dataset")
    raw = raw_clips()
    names = [str(n) for n in g["names"]]
    res = {n: {"audio": raw[n]["audio"], "expression_code": ods.resample(raw[n]["expression_code"], 30, 25),
               "head_orientation": ods.resample(raw[n]["head_orientation"], 30, 25)} for n in names}
    st = coef_stats()
    for mode, rc in (("crop", True), ("nocrop", False)):
        rng = np.random.RandomState(7)
        items = [ods.get_item(res[names[i]], st, rng, random_crop=rc) for i in g[f"{mode}_order"]]
        assert [[it[0][0].shape[0], it[0][1].shape[0]] for it in items] == g[f"{mode}_item_audio_len"].tolist()
        audio, motion, stats = ods.collate(items)
        check_batch(g, mode, audio, motion, stats, 0.0)       # same fp32 arithmetic: bit-exact


@pytest.mark.gpu
def test_resident_dataset_batches_match_reference_and_oracle():
    """One msmd_batch_windows launch per batch reproduces the reference's items + collate bit for bit (copies and one
    fp32 subtract / divide per element), including the short-clip padding branches and the seeded crop draws."""
    import torch
    from msmd_amd.datasets import ResidentDataset
    g = load_golden("g7_dataset")
    raw = raw_clips()
    names = [str(n) for n in g["names"]]
    for mode, rc in (("crop", True), ("nocrop", False)):
        ds = ResidentDataset(raw, names, coef_stats=coef_stats(), original_fps=30, coef_fps=25, random_crop=rc, seed=7)
        audio, coefs, stats = ds.batch(g[f"{mode}_order"])
        torch.cuda.synchronize()
        assert audio[0].shape == (len(g[f"{mode}_order"]), 64000) and coefs[0]["motion"].shape[1:] == (100, 67)
        assert coefs[0]["shape"].shape[1:] == (100, 100) and float(coefs[0]["shape"].abs().sum()) == 0
        check_batch(g, mode, [a.cpu().numpy() for a in audio], [c["motion"].cpu().numpy() for c in coefs],
                    [float(stats[0]), float(stats[1])], 0.0)
    # single item surface + whole-clip query
    ds = ResidentDataset(raw, names, coef_stats=coef_stats(), random_crop=True, seed=7)
    (a0, a1), (c0, c1), (m, s) = ds[0]
    assert [a0.shape[0], a1.shape[0]] == g["crop_item_audio_len"][0].tolist() and c0["motion"].shape == (100, 67)
    audio, cd, _ = ds.query_for_video(3)
    assert audio.shape[0] == raw["short_a"]["audio"].shape[0] and abs(float(audio.mean())) < 1e-4
    # corpus statistics (sum / sum-of-squares over seeded crops) against the reference's incremental_mean_and_std
    ds2 = ResidentDataset(raw, names, coef_stats=None, random_crop=True, seed=11)
    for k in ("exp_mean", "exp_std", "pose_mean", "pose_std"):
        assert np.abs(ds2.coef_stats[k].numpy() - g["stats_" + k]).max() < 2e-5, k


@pytest.mark.gpu
def test_trainer_consumes_resident_dataset_batches():
    """Loader -> Trainer.step hand-off: device-resident batches (no host copy) drive the training iteration."""
    import torch
    from msmd_amd.config import default_args
    from msmd_amd.datasets import ResidentDataset
    from msmd_amd.model import get_diffusion_model
    from msmd_amd.style_encoder import get_style_encoder
    from msmd_amd.training_script import Trainer, batch_from_loader
    raw = raw_clips()
    ds = ResidentDataset(raw, list(raw), coef_stats=None, random_crop=True, seed=3)
    args = default_args(compute_dtype="bf16", encoder_layers=1, n_layers=1, lr=1e-4, warm_iter=0)
    model = get_diffusion_model(args, "cuda").eval()
    se = get_style_encoder(args, "vae2").to("cuda").eval()
    tr = Trainer(args, model, se)
    sampler = np.random.RandomState(0)
    for it in range(1, 4):
        batch = batch_from_loader(ds.batch(sampler.randint(0, len(ds), size=4)))
        assert batch[0][0].shape == (4, 64000) and batch[1][0].shape == (4, 100, 67) and batch[2].shape == (4, 100)
        out = tr.step(batch, it=it)
    torch.cuda.synchronize()
    assert all(bool(torch.isfinite(v).all()) for v in out.values())


@pytest.mark.gpu
def test_train_function_keeps_the_reference_call_signature(tmp_path):
    """training_script.train(args, model, style_enc, train_loader, val_loader, optimizer, save_dir, scheduler, writer,
    ...) as the reference's main() calls it (training_script.py:49-50, 628-629): torch optimizer object and a
    tensorboard-like writer accepted, loader items in the reference's tuple format, checkpoints in its format."""
    import torch
    from msmd_amd.config import default_args
    from msmd_amd.datasets import ResidentDataset
    from msmd_amd.model import get_diffusion_model
    from msmd_amd.style_encoder import get_style_encoder
    from msmd_amd.training_script import count_parameters, train
    raw = raw_clips()
    ds = ResidentDataset(raw, list(raw), coef_stats=None, random_crop=True, seed=5)
    args = default_args(compute_dtype="bf16", encoder_layers=1, n_layers=1, lr=1e-3, warm_iter=0, max_iter=3, save_iter=2,
                        val_iter=3, log_iter=1, batch_size=2)
    model = get_diffusion_model(args, "cuda")
    se = get_style_encoder(args, "vae2").to("cuda")
    assert count_parameters(model) > 0

    class Loader:
        dataset = ds

        def __iter__(self):
            rng = np.random.RandomState(1)
            for _ in range(2):                       # a finite epoch: train() must wrap around it
                yield ds.batch(rng.randint(0, len(ds), size=2))

    class Writer:
        def __init__(self):
            self.tags = []

        def add_scalar(self, tag, value, it):
            assert np.isfinite(value)
            self.tags.append((tag, it))

    opt = torch.optim.Adam([{"params": se.parameters(), "lr": 2e-4}, {"params": model.parameters(), "lr": 2e-4}])
    w = Writer()
    val = [ds.batch([0, 1])]
    tr = train(args, model, se, Loader(), val, opt, tmp_path / "ck", scheduler=None, writer=w, start_iter=0)
    torch.cuda.synchronize()
    assert tr.opt_step == 4 and abs(tr.current_lr() - 2e-4) < 1e-12       # the optimizer's lr seeded the base lr
    assert sorted(p.name for p in (tmp_path / "ck").glob("*.pt")) == ["iter_0000002.pt", "iter_0000003.pt"]
    ck = torch.load(tmp_path / "ck" / "iter_0000003.pt", weights_only=False)
    assert {"args", "model", "style_enc", "iter"} <= set(ck) and ck["iter"] == 3
    tags = {t for t, _ in w.tags}
    assert {"train/loss", "opt/lr", "val/loss"} <= tags


@pytest.mark.gpu
def test_validation_pass_over_resident_batches(tmp_path):
    """training_script.test (reference l.243-403): loss_log keys / aggregation, cross-style rule, model mode restored;
    the per-batch total equals the weighted sum of its parts."""
    import json
    import torch
    from msmd_amd.config import default_args
    from msmd_amd.datasets import ResidentDataset
    from msmd_amd.model import get_diffusion_model
    from msmd_amd.style_encoder import get_style_encoder
    from msmd_amd.training_script import load_loss_weights, test
    raw = raw_clips()
    ds = ResidentDataset(raw, list(raw), coef_stats=None, random_crop=False, seed=3)
    args = default_args(compute_dtype="fp32", encoder_layers=1, n_layers=1)
    model = get_diffusion_model(args, "cuda").train()
    se = get_style_encoder(args, "vae2").to("cuda").eval()
    lw = load_loss_weights(args)
    loader = [ds.batch([2, 3]), ds.batch([4, 5, 3])]
    torch.manual_seed(1)
    log = test(args, lw, model, se, loader, n_rounds=2, coef_stats=ds.coef_stats)
    assert model.training
    assert len(log["loss"]) == 4 and all(np.isfinite(log["loss"]))
    keys = [k for k in lw if lw[k] > 0]
    assert set(log) == set(keys) | {"loss"}
    for j in range(4):
        assert abs(sum(log[k][j] * lw[k] for k in keys) - log["loss"][j]) <= 1e-5 * abs(log["loss"][j])
    path = tmp_path / "metrics.json"
    saved = test(args, lw, model, se, loader, n_rounds=1, do_save=True, do_save_path=path, coef_stats=ds.coef_stats)
    assert json.load(open(path))["loss"]["n_samples"] == 2 and saved["loss"]["n_samples"] == 2


@pytest.mark.gpu
def test_training_cli_end_to_end_on_a_decoded_corpus(tmp_path):
    """training_script.main with the reference's flag names: chunked-pickle corpus -> resident loader -> 4 iterations
    (train-mode noise on) -> checkpoint in the reference's format -> resume -> test mode."""
    import pickle
    import torch
    from msmd_amd.training_script import build_parser, main
    raw = raw_clips()
    # a corpus longer than 2.1 windows after the 30 -> 25 fps resampling, written as two pickle chunks
    names = list(raw)
    path = tmp_path / "corpus.pkl"
    with open(path, "wb") as f:
        pickle.dump({k: raw[k] for k in names[:3]}, f)
        pickle.dump({k: raw[k] for k in names[3:]}, f)
    assert build_parser().parse_args(["--exp_name", "x", "--data_root", "y"]).n_motions == 750   # reference default
    common = ["--exp_name", "cli", "--data_root", str(path), "--exp_root", str(tmp_path), "--n_motions", "100",
              "--n_prev_motions", "10", "--fps", "25", "--audio_model", "wav2vec2", "--rot_repr", "aa", "--use_indicator",
              "--use_cross_style", "--batch_size", "2", "--log_iter", "1", "--warm_iter", "2"]
    tr = main(common + ["--max_iter", "3", "--save_iter", "2", "--val_iter", "3"])
    torch.cuda.synchronize()
    ck = sorted((tmp_path / "cli" / "checkpoints").glob("iter_*.pt"))
    assert [c.name for c in ck] == ["iter_0000002.pt", "iter_0000003.pt"] and (tmp_path / "cli" / "args.json").exists()
    assert tr.opt_step == 4
    tr2 = main(common + ["--max_iter", "5", "--save_iter", "100", "--val_iter", "100", "--continue_from", str(tmp_path / "cli")])
    assert tr2.opt_step == 4 + 3                      # resumed at iteration 3 (inclusive range, as the reference's loop)
    res = main(common + ["--mode", "test", "--max_iter", "5"])
    assert np.isfinite(np.mean(res["loss"]))
    # inference CLI (reference flag names) on the checkpoint just written: style clip pickles + decoded audio -> pickles
    import os
    from msmd_amd import inference
    from msmd_amd.model import DiffusionSchedule
    root = tmp_path / "models"
    os.makedirs(root / "DPT", exist_ok=True)
    os.symlink(tmp_path / "cli", root / "DPT" / "cli")
    stats = {k: torch.from_numpy(v) for k, v in coef_stats().items()}
    files = {}
    for name, obj in (("stats", stats), ("style_exp", torch.from_numpy(raw["long_a"]["expression_code"]).float()),
                      ("style_head", raw["long_a"]["head_orientation"].astype(np.float32))):
        files[name] = tmp_path / f"{name}.pkl"
        with open(files[name], "wb") as f:
            pickle.dump(obj, f)
    np.save(tmp_path / "speech.npy", raw["short_a"]["audio"][:52000])
    real = inference.load_model

    def short_schedule(*a):                              # keep the test fast: 3 denoising steps
        m, se, margs = real(*a)
        m.diffusion_sched = DiffusionSchedule(3, "cosine").to(m.device)
        return m, se, margs
    inference.load_model = short_schedule
    try:
        out = inference.main(["--model_root", str(root), "--model_name", "cli", "--model_iter", "0000003",
                              "--style_clip_exp_code_path", str(files["style_exp"]), "--style_clip_head_rot_path",
                              str(files["style_head"]), "--audio_clip", str(tmp_path / "speech.npy"), "--coef_dict_path",
                              str(files["stats"]), "--output_dir", str(tmp_path / "out"), "--versions_of_render", "2"])
    finally:
        inference.load_model = real
    assert len(out) == 4
    exp = pickle.load(open(out[0], "rb"))
    rot = pickle.load(open(out[1], "rb"))
    assert exp.shape == (int(52000 / 16000 * 25), 64) and rot.shape == (exp.shape[0], 3) and np.isfinite(exp).all()


@pytest.mark.gpu
def test_dataset_pickle_adapter_keeps_reference_constructor(tmp_path):
    """datasets.DatasetPickle(pkl_file, split_file, coef_stats_file, ...) with the reference's arguments
    (datasets.py:167-169): split / valid-id filtering, chunked pickle, stats file, SE items, and get_collate_fn
    reproducing ResidentDataset.batch from an item list."""
    import pickle
    import torch
    from msmd_amd.datasets import DatasetPickle, ResidentDataset
    raw = raw_clips()
    names = list(raw)
    pkl = tmp_path / "corpus.pkl"
    with open(pkl, "wb") as f:
        pickle.dump({k: raw[k] for k in names[:2]}, f)
        pickle.dump({k: raw[k] for k in names[2:]}, f)
    split = tmp_path / "train.txt"
    split.write_text("\n".join(names[:4]) + "\n")
    ds = DatasetPickle(pkl, split, None, celebv_text=False, full_dataset=True, seed=7)
    ref = ResidentDataset(raw, names[:4], seed=7)
    assert ds.file_names == names[:4] and len(ds) == 4
    for k in ("exp_mean", "exp_std", "pose_mean", "pose_std"):
        assert torch.equal(ds.coef_stats[k], ref.coef_stats[k])
    a, b = ds[1], ref[1]
    assert all(torch.equal(x, y) for x, y in zip(a[0], b[0])) and torch.equal(a[1][1]["motion"], b[1][1]["motion"])
    # collate(item list) == the one-launch batch for the same crop draws
    ds2 = DatasetPickle(pkl, split, None, celebv_text=False, full_dataset=True, seed=11)
    ref2 = ResidentDataset(raw, names[:4], coef_stats=ds2.coef_stats, seed=11)
    ref2.rng = np.random.RandomState(5)
    ds2.rng = np.random.RandomState(5)
    got = DatasetPickle.get_collate_fn(False)([ds2[i] for i in (0, 2, 3)])
    want = ref2.batch([0, 2, 3])
    assert torch.equal(got[0][0], want[0][0]) and torch.equal(got[0][1], want[0][1])
    assert torch.equal(got[1][0]["motion"], want[1][0]["motion"]) and torch.equal(got[1][1]["shape"], want[1][1]["shape"])
    assert abs(float(got[2][0]) - float(want[2][0])) < 1e-7 and abs(float(got[2][1]) - float(want[2][1])) < 1e-7
    # stats file + style-encoder items + valid-id filter + over-fit truncation
    np.savez(tmp_path / "stats.npz", **{k: v.numpy() for k, v in ds.coef_stats.items()})
    keys = tmp_path / "keys.txt"
    keys.write_text("\n".join([names[0], names[3], "not_in_split"]) + "\n")
    old = DatasetPickle.VALID_ID_FILE
    DatasetPickle.VALID_ID_FILE = str(keys)
    try:
        se = DatasetPickle(pkl, split, tmp_path / "stats.npz", SE=True, celebv_text=True, full_dataset=True, seed=3)
    finally:
        DatasetPickle.VALID_ID_FILE = old
    assert se.file_names == [names[0], names[3]]
    item = se[0]
    assert len(item) == 2 and item[0].shape == (100, 67)
    pair = DatasetPickle.get_collate_fn(True)([se[0], se[1]])
    assert pair[0].shape == (2, 100, 67) and pair[1].shape == (2, 100, 67)
    one = DatasetPickle(pkl, split, tmp_path / "stats.npz", celebv_text=False, pre_loaded_raw_dataset=raw,
                        batch_overfit_size=1)
    assert len(one) == 1
    assert len(list(DatasetPickle.load_dict_in_chunks_static(pkl))) == 2


@pytest.mark.gpu
def test_training_cli_from_a_local_pretrained_encoder_keeps_it(tmp_path, monkeypatch):
    """reference training_script.py:527-528 + model.py:95-101: a fresh run builds the model around the PRETRAINED audio
    encoder.  training_script.main on a local Hugging Face checkpoint directory (real widths, one transformer layer; no
    synthetic switch anywhere): after two iterations the frozen feature extractor in the written checkpoint still equals the
    file bit for bit, the trainable encoder tensors are the file's plus two small Adam steps, the other parameters came from
    the reference's initialisation (not the closed-form fill)."""
    import json
    import os
    import pickle
    import torch
    from safetensors.torch import save_file
    from msmd_amd import shapes
    from msmd_amd.training_script import main
    monkeypatch.delenv("MSMD_SYNTHETIC_WEIGHTS", raising=False)
    ck = tmp_path / "w2v"
    os.makedirs(ck)
    cfg = dict(model_type="wav2vec2", num_hidden_layers=1, hidden_size=768, intermediate_size=3072, num_attention_heads=12,
               conv_dim=[512] * 7, conv_kernel=[10, 3, 3, 3, 3, 2, 2], conv_stride=[5, 2, 2, 2, 2, 2, 2],
               num_conv_pos_embeddings=128, num_conv_pos_embedding_groups=16, layer_norm_eps=1e-5, feat_extract_norm="group",
               conv_bias=False, do_stable_layer_norm=False, architectures=["Wav2Vec2ForCTC"])
    with open(ck / "config.json", "w") as f:
        json.dump(cfg, f)
    file_sd = {"wav2vec2." + k: torch.from_numpy(synth.fill_tensor("pretrained." + k, s)).contiguous()
               for k, s in shapes.audio_encoder_shapes(1).items()}
    save_file(file_sd, str(ck / "model.safetensors"))
    raw = raw_clips()
    path = tmp_path / "corpus.pkl"
    with open(path, "wb") as f:
        pickle.dump(raw, f)
    torch.manual_seed(11)
    tr = main(["--exp_name", "pre", "--data_root", str(path), "--exp_root", str(tmp_path), "--n_motions", "100",
               "--n_prev_motions", "10", "--fps", "25", "--audio_model", "wav2vec2", "--rot_repr", "aa", "--use_indicator",
               "--batch_size", "2", "--log_iter", "1", "--warm_iter", "0", "--lr", "1e-5", "--max_iter", "1", "--save_iter", "1",
               "--val_iter", "100", "--audio_encoder_weights", str(ck)])
    torch.cuda.synchronize()
    assert tr.model.audio_encoder.weights_source == str(ck) and tr.opt_step == 2
    saved = torch.load(tmp_path / "pre" / "checkpoints" / "iter_0000001.pt", weights_only=False)["model"]
    n_frozen = n_trained = 0
    for k, want in file_sd.items():
        k = "audio_encoder." + k[len("wav2vec2."):]
        got = saved[k].cpu().float()
        if k.startswith("audio_encoder.feature_extractor."):
            assert torch.equal(got, want), k
            n_frozen += 1
        elif "masked_spec_embed" not in k:
            assert float((got - want).abs().max()) <= 2 * 1e-5 * 1.01 + 1e-7, k     # two Adam steps of at most lr each
            n_trained += 1
    assert n_frozen == 9 and n_trained >= 20
    # the rest is the reference's initialisation, not the closed-form fill, and follows the caller's generator
    w = saved["denoising_net.person_proj.weight"].cpu().float()
    assert not torch.allclose(w, torch.from_numpy(synth.fill_tensor("denoising_net.person_proj.weight", w.shape)), atol=1e-3)
    assert abs(float(w.std()) - (1.0 / 356 ** 0.5) / 3 ** 0.5) < 0.05 * float(w.std())      # U(+-1/sqrt fan_in)
