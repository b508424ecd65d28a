"""Checkpoint compatibility with disvae/utils/modelIO.py:14-42,81-153: same file names,
same state_dict keys/shapes, so reference checkpoints (results/*/model.pt) load into the
native model and vice versa."""
import json
import os

import torch

MODEL_FILENAME = "model.pt"
META_FILENAME = "specs.json"


def save_metadata(metadata, directory, filename=META_FILENAME, **kwargs):
    with open(os.path.join(directory, filename), "w") as f:
        json.dump(metadata, f, indent=4, sort_keys=True, **kwargs)


def load_metadata(directory, filename=META_FILENAME):
    with open(os.path.join(directory, filename)) as f:
        return json.load(f)


def save_model(model, directory, metadata=None, filename=MODEL_FILENAME):
    if metadata is None:
        metadata = dict(img_size=model.img_size, latent_dim=model.latent_dim, model_type=model.model_type)
    os.makedirs(directory, exist_ok=True)
    save_metadata(metadata, directory)
    state = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    torch.save(state, os.path.join(directory, filename))


def load_model(directory, is_gpu=True, filename=MODEL_FILENAME):
    from ..models.vae import init_specific_model
    device = torch.device("cuda" if torch.cuda.is_available() and is_gpu else "cpu")
    meta = load_metadata(directory)
    model = init_specific_model(meta["model_type"], tuple(meta["img_size"]), meta["latent_dim"])
    state = torch.load(os.path.join(directory, filename), map_location="cpu")
    model.load_state_dict(state, strict=False)
    model = model.to(device)
    model.eval()
    return model
