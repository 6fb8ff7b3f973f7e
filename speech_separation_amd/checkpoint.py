"""Checkpoint reload with the reference's conventions
(reference: utils/misc.py:45-52 ``load_checkpoint``, :59-97 ``reload_for_eval``).

A checkpoint directory holds a text pointer file ``last_best_checkpoint``
(preferred) or ``last_checkpoint`` whose first line names the checkpoint file;
the file is a dict with the weights under ``'model'`` (or the bare
state_dict).  Tensors are taken when name (also with / without a ``module.``
prefix) and shape match; others are reported and left at their current value.
Problems are printed, not raised, as in the reference.
"""
import os

import torch


def load_checkpoint(checkpoint_path, use_cuda):
    print("Inferencing - Using GPU" if use_cuda else "Inferencing - Using CPU")
    try:
        return torch.load(checkpoint_path, map_location="cpu", weights_only=True)
    except Exception as e:  # training checkpoints carry optimizer / RNG state that is not plain tensors
        print(f"safe load refused {checkpoint_path} ({type(e).__name__}); loading it as a trusted training checkpoint")
        return torch.load(checkpoint_path, map_location="cpu", weights_only=False)


def reload_for_eval(model, checkpoint_dir, use_cuda):
    print("reloading from: {}".format(checkpoint_dir))
    pointer = None
    for name in ("last_best_checkpoint", "last_checkpoint"):
        cand = os.path.join(checkpoint_dir, name)
        if os.path.isfile(cand):
            pointer = cand
            break
    if pointer is None:
        print("Warning: There is no exited checkpoint or best_model!!!!!!!!!!!!")
        return
    with open(pointer, "r") as f:
        model_name = f.readline().strip()
    checkpoint_path = os.path.join(checkpoint_dir, model_name)
    print("checkpoint_path: {}".format(checkpoint_path))
    checkpoint = load_checkpoint(checkpoint_path, use_cuda)
    pretrained = checkpoint["model"] if "model" in checkpoint else checkpoint
    state = model.state_dict()
    for key in state.keys():
        for alias in (key, key.replace("module.", ""), "module." + key):
            if alias in pretrained and state[key].shape == pretrained[alias].shape:
                state[key] = pretrained[alias]
                break
        else:
            print(f"{key} not loaded")
    model.load_state_dict(state)
    print("=> Reload well-trained model {} for decoding.".format(model_name))
