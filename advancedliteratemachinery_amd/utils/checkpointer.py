"""Checkpoint loading with the reference's file format (OCR/OmniParser/utils/checkpointer.py:19-65):
a torch.save pickle holding {'model': state_dict, ...} or a bare state_dict; with --train_vie the
vocabulary rows of 7 tensors grow by `vie_categories` (old rows copied, new rows kept)."""
import torch

_VOCAB_KEYS = ['transformer.embedding.word_embeddings.weight'] + [
    'transformer.%s_pred_layer.layers.2.%s' % (k, leaf) for k in ('pt', 'poly', 'rec') for leaf in ('weight', 'bias')]


class Checkpointer(object):
    def __init__(self, distributed=False):
        self.distributed = distributed

    def load(self, checkpoint_path, model, args, optimizer=None, lr_scheduler=None):
        ckpt = torch.load(checkpoint_path, map_location='cpu', weights_only=False)
        sd = ckpt['model'] if isinstance(ckpt, dict) and 'model' in ckpt else ckpt
        target = model.module if hasattr(model, 'module') else model
        if getattr(args, 'train_vie', False) and not getattr(args, 'continue_train', False):
            new = target.state_dict()
            for k in new:
                if k in _VOCAB_KEYS:
                    new[k][:-args.vie_categories] = sd[k]
                else:
                    new[k] = sd[k]
            sd = new
        target.load_state_dict(sd)
        epoch = ckpt.get('epoch', -1) if isinstance(ckpt, dict) else -1
        step = ckpt.get('global_step', 0) if isinstance(ckpt, dict) else 0
        return epoch, step
