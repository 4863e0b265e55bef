"""Config / command line with the reference's keys (next row N4): merge order and coercion on CPU, one
train + predict round trip through the command line on the GPU."""
import argparse
import os

import numpy as np
import pytest
import yaml


def test_defaults_and_merge_order(tmp_path):
    from qfa_amd import config as Cf
    from qfa_amd.cli import build_parser
    c = Cf.get_config()
    # the reference's defaults (QFA/config.py:15-63)
    assert (c.TYPE, c.GPU, c.DATA.BATCH_SIZE, c.DATA.DATA_NUM, c.DATA.LAMMIN, c.DATA.LAMMAX, c.DATA.LOGLAM_DELTA) == \
        ("train", 0, 500, 10000, 1030.0, 1600.0, 1e-4)
    assert (c.MODEL.NH, c.MODEL.TAU, c.TRAIN.NEPOCHS, c.TRAIN.LEARNING_RATE, c.TRAIN.WEIGHT_DECAY,
            c.TRAIN.DECAY_ALPHA, c.TRAIN.DECAY_STEP, c.TRAIN.WINDOW_LENGTH_FOR_MU) == (8, "becker", 500, 1e-3, 1e-1, 0.9, 10, 16)
    base = tmp_path / "base.yaml"
    base.write_text(yaml.safe_dump({"MODEL": {"NH": 12, "TAU": "fg"}, "TRAIN": {"NEPOCHS": 7}}))
    top = tmp_path / "top.yaml"
    top.write_text(yaml.safe_dump({"BASE": ["base.yaml"], "MODEL": {"NH": 16}, "DATA": {"BATCH_SIZE": 64}}))
    args = build_parser().parse_args(["--cfg", str(top), "--opts", "TRAIN.LEARNING_RATE", "5e-3", "DATA.Z_MAX", "3",
                                      "--n_epochs", "3", "--tau", "kamble", "--type", "predict", "--Nh", "4"])
    c = Cf.get_config(args)
    assert c.MODEL.NH == 16 and c.MODEL.TAU == "kamble"         # flags win over files; --Nh is ignored (main.py:25)
    assert c.TRAIN.NEPOCHS == 3 and c.DATA.BATCH_SIZE == 64 and c.TYPE == "predict"
    assert c.TRAIN.LEARNING_RATE == 5e-3 and c.DATA.Z_MAX == 3.0 and isinstance(c.DATA.Z_MAX, float)
    again = yaml.safe_load(c.dump())
    assert again["MODEL"]["NH"] == 16 and again["DATA"]["BATCH_SIZE"] == 64
    bad = tmp_path / "bad.yaml"
    bad.write_text(yaml.safe_dump({"MODEL": {"NOPE": 1}}))
    with pytest.raises(KeyError):
        Cf.get_config(argparse.Namespace(cfg=str(bad), opts=None))
    with pytest.raises(KeyError):
        Cf.get_config(argparse.Namespace(cfg=None, opts=["DATA.NOPE", "1"]))
    with pytest.raises(ValueError):
        Cf.get_config(argparse.Namespace(cfg=None, opts=["DATA.Z_MAX"]))


def test_schema_pinned_to_the_reference():
    """DEFAULTS / ARG_KEYS / the parser's flags against tests/golden/g12_config.json, which
    tests/golden/make_config_golden.py extracts from the reference's QFA/config.py:15-63,92-139 and main.py:16-42."""
    import json
    from conftest import GOLDEN
    from qfa_amd import config as Cf
    from qfa_amd.cli import build_parser
    g = json.load(open(os.path.join(GOLDEN, "g12_config.json")))

    def flat(d, pre=""):
        out = {}
        for k, v in d.items():
            if isinstance(v, dict):
                out.update(flat(v, pre + k + "."))
            else:
                out[pre + k] = v
        return out
    ours = flat(Cf.DEFAULTS)
    for k in Cf.EXTRA_KEYS:                                      # documented additions, nothing else
        assert k in ours and k not in g["defaults"]
        del ours[k]
    assert ours == g["defaults"]
    for k, v in g["defaults"].items():                           # same types too (yacs refuses type changes)
        assert type(ours[k]) is type(v), k
    assert Cf.ARG_KEYS == g["arg_keys"]
    types = {"int": int, "str": str, "float": float, "bool": bool, None: None}
    acts = {a.dest: a for a in build_parser()._actions if a.dest != "help"}
    # --nprocs: QFA/config.py:130 reads args.nprocs but main.py never defines the flag; here it exists
    assert set(acts) - {"nprocs"} == set(g["flags"])
    for name, spec in g["flags"].items():
        if name == "opts":          # the reference's --opts takes ONE string, which merge_from_list cannot use; here KEY VALUE pairs
            continue
        assert acts[name].type is types[spec["type"]], name
        assert acts[name].nargs == spec["nargs"], name
        assert not acts[name].required


def test_parser_has_the_reference_flags():
    from qfa_amd.cli import build_parser
    flags = {a.dest for a in build_parser()._actions}
    for f in ("cfg", "catalog", "type", "data_num", "validation_catalog", "validation_num", "batch_size", "n_epochs", "Nh",
              "tau", "learning_rate", "gpu", "snr_min", "snr_max", "z_min", "z_max", "num_mask", "decay_alpha",
              "decay_step", "weight_decay", "output_dir", "data_dir", "validation_dir", "validation", "opts"):
        assert f in flags, f


@pytest.mark.gpu
def test_cli_train_then_predict(tmp_path):
    import pandas as pd
    from qfa_amd import cli, io, synthetic
    lam = dict(LAMMIN=1030.0, LAMMAX=1600.0, LOGLAM_DELTA=2e-3)
    wav = io.wavelength_grid(lam["LAMMIN"], lam["LAMMAX"], lam["LOGLAM_DELTA"])
    npix, nb = len(wav), int(np.sum(wav < 1215.67))
    p, mu = synthetic.mock_parameters(npix, nb, 4, seed=9)
    b = synthetic.make_batch_numpy(p, mu, wav, nb, 24, seed=91, masks=False)
    data = tmp_path / "data"
    data.mkdir()
    names = []
    for i in range(24):
        names.append(f"spec-{i:02d}.npz")
        np.savez(data / names[-1], flux=b["flux"][i].astype(np.float64), error=b["error"][i].astype(np.float64), z=b["zqso"][i])
    pd.DataFrame({"file": names, "snr": 5.0, "z": b["zqso"], "num_mask": 0}).to_csv(tmp_path / "catalog.csv", index=False)
    cfgf = tmp_path / "cfg.yaml"
    cfgf.write_text(yaml.safe_dump({"DATA": dict(lam, DATA_DIR=str(data), CATALOG=str(tmp_path / "catalog.csv"),
                                                 OUTPUT_DIR=str(tmp_path / "out"), DATA_NUM=24, BATCH_SIZE=8, NPROCS=2,
                                                 Z_MIN=0.0, Z_MAX=10.0),
                                    "MODEL": {"NH": 4, "TAU": "becker"}, "TRAIN": {"NEPOCHS": 5}}))
    np.random.seed(4)
    assert cli.main(["--cfg", str(cfgf), "--type", "train"]) == 0
    out = tmp_path / "out"
    assert (out / "config.yaml").exists() and (out / "log.txt").exists() and (out / "train-catalog.csv").exists()
    ck = out / "checkpoints" / "model_parameters_epoch_05.npz"
    assert ck.exists()
    assert "epoch: 004/005" in (out / "log.txt").read_text()
    pd.Series(names[:5]).to_csv(tmp_path / "pred.csv", header=False, index=False)
    # the reference's predict catalogue is read with a header row: the first line names the column
    (tmp_path / "pred.csv").write_text("file\n" + "\n".join(names[:5]) + "\n")
    assert cli.main(["--cfg", str(cfgf), "--type", "predict", "--catalog", str(tmp_path / "pred.csv"),
                     "--opts", "MODEL.RESUME", str(ck)]) == 0
    for n in names[:5]:
        r = np.load(out / "predict" / n)
        assert r["ll"].shape == (1, 1) and r["hmean"].shape == (4, 1) and r["hcov"].shape == (4, 4)
        assert r["cont"].shape == (npix,) and r["uncertainty"].shape == (npix,)
        assert np.isfinite(r["cont"]).all() and np.isfinite(r["ll"]).all()
    # one predicted file against the float64 oracle on the checkpoint the command line itself wrote
    # (load_from_npz semantics: c0 <- beta, quirk Q1; zabs / mask / delta as the reference's Dataloader builds them)
    from oracle import qfa_oracle as O
    pp, pmu = O.load_params_npz(str(ck))
    for i in (0, 4):
        f64, e64 = b["flux"][i].astype(np.float64), b["error"][i].astype(np.float64)
        zabs = O.zabs_from_zqso(wav, np.asarray([b["zqso"][i]], dtype=np.float64), nb)[0]
        mask = (f64 != -999.0) & (e64 != -999.0)
        o = O.predict_single(pp, pmu, f64.astype(np.float32), e64.astype(np.float32), zabs.astype(np.float32), mask)
        r = np.load(out / "predict" / names[i])
        assert abs(r["ll"].item() - o[0]) / abs(o[0]) < 1e-5
        assert np.max(np.abs(r["cont"] - o[3])) / np.max(np.abs(o[3])) < 1e-4
        assert np.linalg.norm(r["hmean"].ravel() - o[1]) / np.linalg.norm(o[1]) < 1e-4
        assert np.linalg.norm(r["uncertainty"] - o[4]) / np.linalg.norm(o[4]) < 1e-4
