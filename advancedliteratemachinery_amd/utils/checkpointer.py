"""Checkpoint loading with the reference's file format (OCR/OmniParser/utils/checkpointer.py:19-65):
a torch.save pickle holding {'model': state_dict, 'optimizer', 'lr_scheduler', 'epoch', 'global_step', 'args'} (step
checkpoints), {'model': state_dict} (the final one) or a bare state_dict; with --train_vie the vocabulary rows of 7
tensors grow by `vie_categories` (old rows copied, new rows kept).

Safety: a checkpoint path comes from the command line (--resume).  The file is first read with
torch.load(weights_only=True) with argparse.Namespace allow-listed (the only non-tensor object the reference stores,
under 'args'); full unpickling -- which can execute code -- happens only with `allow_unsafe_pickle=True`.
"""
import argparse
import pickle

import torch

_VOCAB_KEYS = ['transformer.embedding.word_embeddings.weight'] + [
    'transformer.%s_pred_layer.layers.2.%s' % (k, leaf) for k in ('pt', 'poly', 'rec') for leaf in ('weight', 'bias')]


def load_checkpoint_file(path, allow_unsafe_pickle=False):
    """I/O errors (missing or truncated file, unsupported format) propagate as they are; only the weights-only unpickler's
    refusal of a non-allow-listed object is turned into the "pass allow_unsafe_pickle" error -- or, with that flag, into a full
    (code-executing) unpickle of a file the caller vouches for."""
    try:
        with torch.serialization.safe_globals([argparse.Namespace]):
            return torch.load(path, map_location='cpu', weights_only=True)
    except pickle.UnpicklingError as e:
        if 'weights only' not in str(e).lower() and 'weights_only' not in str(e).lower():
            raise    # a corrupt pickle stream, not an allow-list refusal
        if not allow_unsafe_pickle:
            raise RuntimeError('%s holds objects torch.load(weights_only=True) refuses (%s); pass allow_unsafe_pickle=True '
                               '(--allow_unsafe_pickle) only for files you trust' % (path, str(e).splitlines()[0])) from e
        return torch.load(path, map_location='cpu', weights_only=False)


class Checkpointer(object):
    def __init__(self, distributed=False, allow_unsafe_pickle=False):
        self.distributed = distributed
        self.allow_unsafe_pickle = allow_unsafe_pickle

    def load(self, checkpoint_path, model, args, optimizer=None, lr_scheduler=None):
        """-> (last_epoch, global_step), exactly as the reference: anything but the weights is dropped unless
        args.continue_train (checkpointer.py:22-25), so a fresh fine-tune / eval starts at (-1, 0)."""
        ckpt = load_checkpoint_file(checkpoint_path, self.allow_unsafe_pickle)
        cont = bool(getattr(args, 'continue_train', False))
        has_model = isinstance(ckpt, dict) and 'model' in ckpt and isinstance(ckpt['model'], dict)
        if not cont:
            if not has_model:
                # the reference indexes checkpoint['model'] here (KeyError on a bare state-dict); a bare dict is accepted
                ckpt = {'model': ckpt}
            else:
                ckpt = {'model': ckpt['model']}
            has_model = True
        sd = ckpt['model'] if has_model else ckpt
        target = model.module if (self.distributed or hasattr(model, 'module')) and hasattr(model, 'module') else model
        if getattr(args, 'train_vie', False) and not cont:
            new = target.state_dict()
            for k in new:
                if k in _VOCAB_KEYS:
                    new[k][:-args.vie_categories] = sd[k]
                else:
                    new[k] = sd[k]
            sd = new
        target.load_state_dict(sd)
        if optimizer is not None and has_model and 'optimizer' in ckpt:
            optimizer.load_state_dict(ckpt['optimizer'])
        if lr_scheduler is not None and has_model and 'lr_scheduler' in ckpt:
            lr_scheduler.load_state_dict(ckpt['lr_scheduler'])
        epoch = ckpt.get('epoch', -1) if has_model else -1
        step = ckpt.get('global_step', 0) if has_model else 0
        return epoch, step
