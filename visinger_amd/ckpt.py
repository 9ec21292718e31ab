"""Checkpoint reader / writer compatible with the reference's files.

Reference format (``utils/commons/trainer.py:473-492``): ``torch.save`` with the LEGACY (non-zip) serialisation of
``{'epoch', 'global_step', 'checkpoint_callback_best', 'optimizer_states': [...], 'state_dict': {child_name:
child.state_dict()}}`` where the task's children are ``model`` (the generator-side VISinger) and ``mel_disc``
(MultiPeriodDiscriminator); written to ``<path>.part`` then ``os.replace``d; named ``model_ckpt_steps_{N}.ckpt`` and
resumed newest-first by the step number in the file name (``utils/commons/ckpt_utils.py:8-25``)."""
import glob
import os
import re

import torch


def all_checkpoints(work_dir, steps=None):
    """newest first, by the step number in the file name (ckpt_utils.py:17-25)"""
    pat = f"{work_dir}/model_ckpt_steps_{'*' if steps is None else steps}.ckpt"
    return sorted(glob.glob(pat), key=lambda p: -int(re.findall(r".*steps_(\d+)\.ckpt", p)[0]))


def read_checkpoint(path_or_dir):
    """-> (checkpoint dict, path).  `path_or_dir` is a .ckpt/.pt file or a directory (newest checkpoint is taken)."""
    if os.path.isdir(path_or_dir):
        paths = all_checkpoints(path_or_dir)
        if not paths:
            raise FileNotFoundError(f"no model_ckpt_steps_*.ckpt in {path_or_dir}")
        path_or_dir = paths[0]
    return torch.load(path_or_dir, map_location="cpu", weights_only=False), path_or_dir


def load_model(model, path_or_dir, child="model", strict=True):
    """Load ``checkpoint['state_dict'][child]`` into `model` (ckpt_utils.py:28-56).  The parameter names and shapes of
    visinger_amd's VISinger / MultiPeriodDiscriminator are the reference's, so strict=True works on its files."""
    ckpt, path = read_checkpoint(path_or_dir)
    model.load_state_dict(ckpt["state_dict"][child], strict=strict)
    return ckpt.get("global_step", 0), path


def save_checkpoint(path, children, optimizers=(), epoch=0, global_step=0, best=float("inf")):
    """Write a reference-format checkpoint atomically (trainer.py:473-492).  children: {'model': VISinger,
    'mel_disc': MultiPeriodDiscriminator}."""
    ckpt = {"epoch": epoch, "global_step": global_step, "checkpoint_callback_best": best,
            "optimizer_states": [o.state_dict() for o in optimizers if o is not None],
            "state_dict": {k: {n: t.detach().cpu() for n, t in m.state_dict().items()} for k, m in children.items()
                           if len(list(m.parameters())) > 0}}
    tmp = str(path) + ".part"
    torch.save(ckpt, tmp, _use_new_zipfile_serialization=False)
    os.replace(tmp, path)
    return path
