"""Command line of the reference's main.py on the MI355X path.

    python -m qfa_amd.cli --cfg config.yaml --type train  [--catalog ... --data_dir ... --output_dir ...]
    python -m qfa_amd.cli --cfg config.yaml --type predict

Same flags, config keys and outputs as reference main.py:16-101: ``config.yaml`` and ``log.txt`` in OUTPUT_DIR,
checkpoints every five epochs, ``predict/<name>.npz`` with ll, hmean, hcov, cont, uncertainty.  Spectra are read
on the host (qfa_amd.io), everything after that runs on the GPU; there is no CPU device.
"""
from __future__ import annotations

import argparse
import logging
import os
import time
from functools import partial


def build_parser():
    p = argparse.ArgumentParser(description="QFA training / prediction on MI355X")
    p.add_argument("--cfg", type=str)
    p.add_argument("--catalog", type=str)
    p.add_argument("--type", type=str, help="train or predict")
    p.add_argument("--data_num", type=int)
    p.add_argument("--validation_catalog", type=str)
    p.add_argument("--validation_num", type=int)
    p.add_argument("--batch_size", type=int)
    p.add_argument("--n_epochs", type=int)
    p.add_argument("--Nh", type=int)
    p.add_argument("--tau", type=str)
    p.add_argument("--learning_rate", type=float)
    p.add_argument("--gpu", type=int)
    p.add_argument("--snr_min", type=float)
    p.add_argument("--snr_max", type=float)
    p.add_argument("--z_min", type=float)
    p.add_argument("--z_max", type=float)
    p.add_argument("--num_mask", type=int)
    p.add_argument("--decay_alpha", type=float)
    p.add_argument("--decay_step", type=int)
    p.add_argument("--weight_decay", type=float)
    p.add_argument("--output_dir", type=str)
    p.add_argument("--data_dir", type=str)
    p.add_argument("--validation_dir", type=str)
    p.add_argument("--validation", type=bool)
    p.add_argument("--nprocs", type=int)
    p.add_argument("--opts", nargs="*", help="KEY VALUE pairs, e.g. MODEL.NH 16")
    return p


def load_data(cfg, device):
    """the reference Dataloader constructor (QFA/dataloader.py:58-112) on qfa_amd.io + DeviceDataloader"""
    import numpy as np
    from . import io
    from .dataloader import DeviceDataloader
    D = cfg.DATA
    wav = io.wavelength_grid(D.LAMMIN, D.LAMMAX, D.LOGLAM_DELTA)
    if cfg.TYPE == "train":
        parts = [io.load_from_catalog(D.CATALOG, D.DATA_DIR, D.DATA_NUM, D.SNR_MIN, D.SNR_MAX, D.Z_MIN, D.Z_MAX,
                                      D.NUM_MASK, D.NPROCS, D.OUTPUT_DIR, "train")]
        if D.VALIDATION and os.path.exists(D.VALIDATION_CATALOG) and os.path.exists(D.VALIDATION_DIR):
            parts.append(io.load_from_catalog(D.VALIDATION_CATALOG, D.VALIDATION_DIR, D.VALIDATION_NUM, D.SNR_MIN,
                                              D.SNR_MAX, D.Z_MIN, D.Z_MAX, D.NUM_MASK, D.NPROCS, D.OUTPUT_DIR,
                                              "validation"))
        flux, error, zqso, paths = [np.concatenate([p[i] for p in parts]) for i in range(4)]
    elif cfg.TYPE == "predict":
        files = io.read_prediction_catalog(D.CATALOG)
        flux, error, zqso, paths = io.read_spectra([os.path.join(D.DATA_DIR, x) for x in files], D.NPROCS)
    else:
        raise NotImplementedError("TYPE should be in ['train', 'predict']!")
    return DeviceDataloader(flux, error, zqso, wav, D.BATCH_SIZE, device, tau=cfg.MODEL.TAU,
                            window_length_for_mu=cfg.TRAIN.WINDOW_LENGTH_FOR_MU, mode=cfg.TYPE, paths=paths)


def load_model_file(model, path, cfg, optimizer=None):
    """A file written by QFA.save_checkpoint (it carries adam_* keys and the true c0) goes through load_checkpoint;
    anything else through the reference's loader, with its c0 <- beta quirk unless MODEL.REFERENCE_C0_QUIRK is off.
    Returns True when the optimiser state was restored too."""
    import numpy as np
    with np.load(path) as f:
        full = "adam_i" in f.files
    if full:
        model.load_checkpoint(path, optimizer)
        return optimizer is not None
    model.load_from_npz(path, reference_c0_quirk=bool(cfg.MODEL.REFERENCE_C0_QUIRK))
    return False


def main(argv=None):
    from .config import get_config
    args = build_parser().parse_args(argv)
    cfg = get_config(args)
    assert cfg.TYPE in ("train", "predict"), "TYPE must be in ['train', 'predict']!"
    os.makedirs(cfg.DATA.OUTPUT_DIR, exist_ok=True)
    with open(os.path.join(cfg.DATA.OUTPUT_DIR, "config.yaml"), "w") as f:
        f.write(cfg.dump())

    import torch
    from . import QFA, Adam, step_scheduler
    from .utils import tau as taufunc
    device = torch.device("cuda", int(cfg.GPU))
    torch.cuda.set_device(device)
    dataloader = load_data(cfg, device)
    model = QFA(dataloader.Nb, dataloader.Nr, cfg.MODEL.NH, device=device, tau=partial(taufunc, which=cfg.MODEL.TAU))
    if cfg.TYPE == "train":
        logger = logging.getLogger("qfa_amd")
        logger.setLevel(logging.INFO)
        handler = logging.FileHandler(os.path.join(cfg.DATA.OUTPUT_DIR, "log.txt"))
        handler.setFormatter(logging.Formatter("%(asctime)s - %(name)s - %(levelname)s - %(message)s"))
        logger.addHandler(handler)
        scheduler = step_scheduler(cfg.TRAIN.DECAY_ALPHA, cfg.TRAIN.DECAY_STEP)
        optimizer = Adam(params=model.parameters, learning_rate=cfg.TRAIN.LEARNING_RATE, device=device,
                         scheduler=scheduler, weight_decay=cfg.TRAIN.WEIGHT_DECAY)
        full_state = False
        if cfg.MODEL.RESUME and os.path.exists(cfg.MODEL.RESUME):
            print(f"=> Resume from {cfg.MODEL.RESUME}")
            full_state = load_model_file(model, cfg.MODEL.RESUME, cfg, optimizer)
        if not full_state:
            # as main.py:83: the reference re-randomises even after a resume (quirk Q8); a checkpoint of this
            # package that carries the Adam state (QFA.save_checkpoint) continues instead
            model.random_init_func()
        model.train(optimizer, dataloader, cfg.TRAIN.NEPOCHS, cfg.DATA.OUTPUT_DIR, logger=logger)
    else:
        print(f"try to predict {len(dataloader)} spectra...")
        print(f"=> Resume from {cfg.MODEL.RESUME}")
        load_model_file(model, cfg.MODEL.RESUME, cfg)         # parameters and mu of the trained model
        ts = time.time()
        model.predict_to_npz(dataloader, os.path.join(cfg.DATA.OUTPUT_DIR, "predict"))
        print(f"Finish predicting {len(dataloader)} spectra in {time.time() - ts} seconds...")
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
