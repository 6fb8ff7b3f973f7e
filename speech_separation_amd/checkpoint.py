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
import pickle

import torch


def load_checkpoint(checkpoint_path, use_cuda, trust=None):
    """Reads the checkpoint with the loader that executes nothing from the file
    (``weights_only=True``): tensors, containers and numbers, which is what the
    reference's trainer writes (reference: solver.py:222-252).  A file that
    loader refuses is NOT unpickled unless the caller opts in -- ``trust=True``
    or ``MSF2_TRUST_CHECKPOINT=1`` in the environment -- because unpickling runs
    code from the file (the reference itself always unpickles,
    utils/misc.py:48-51)."""
    print("Inferencing - Using GPU" if use_cuda else "Inferencing - Using CPU")
    if trust is None:
        trust = os.environ.get("MSF2_TRUST_CHECKPOINT") == "1"
    try:
        return torch.load(checkpoint_path, map_location="cpu", weights_only=True)
    except pickle.UnpicklingError as e:
        if not trust:
            print(f"safe load refused {checkpoint_path}: {e}\n"
                  "set MSF2_TRUST_CHECKPOINT=1 (or trust=True) to unpickle a checkpoint you trust")
            raise
        print(f"safe load refused {checkpoint_path}; unpickling it as a trusted checkpoint")
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
