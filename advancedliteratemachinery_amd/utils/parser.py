"""Config surface of the OmniParser hot path.

Mirrors the reference's single argparse class so that the same command lines
(`--tfm_pre_norm --use_fpn --use_char_window_prompt --infer_vie ...`) drive this
engine.  Reference: OCR/OmniParser/utils/parser.py:3-105 (flag names/defaults) and
:88-105 (derived vocabulary indices).

Vocabulary layout derived in `finalize_args` (reference parser.py:91-103):

    [0, num_bins)                      coordinate bins
    [num_bins, num_bins+len(chars)]    characters + 'unknown'
    recog_pad, pt_eos, poly_eos, rec_eos, pt_sos, poly_sos, rec_sos, padding
    [padding+1, padding+1+vie_categories)   KIE class tokens
"""
import argparse

DEFAULT_CHARS = (' !"#$%&\'()*+,-./0123456789:;<=>?@ABCDEFGHIJKLMNOPQRSTUVWXYZ'
                 '[\\]^_`abcdefghijklmnopqrstuvwxyz{|}~')

# (flag, kwargs) table; kept declarative so tests can diff it against the reference parser.
_FLAGS = [
    # data
    ('--data_root', dict(type=str)),
    ('--train_dataset', dict(type=str, nargs='+')),
    ('--val_dataset', dict(type=str, nargs='+')),
    ('--train_min_size', dict(type=int, nargs='+',
                              default=[640, 672, 704, 736, 768, 800, 832, 864, 896])),
    ('--train_max_size', dict(type=int, default=1920)),
    ('--test_min_size', dict(type=int, default=1200)),
    ('--test_max_size', dict(type=int, default=1920)),
    ('--chars', dict(type=str, default=DEFAULT_CHARS)),
    # sequence construction
    ('--num_bins', dict(type=int, default=1000)),
    ('--rec_length', dict(type=int, default=25)),
    ('--pt_seq_length', dict(type=int, default=1024)),
    # augmentation (accepted for command-line compatibility; unused at inference)
    ('--crop_min_size_ratio', dict(type=float, default=0.3)),
    ('--crop_max_size_ratio', dict(type=float, default=1.0)),
    ('--crop_prob', dict(type=float, default=1.0)),
    ('--rotate_max_angle', dict(type=int, default=90)),
    ('--rotate_prob', dict(type=float, default=0.5)),
    ('--dist_brightness', dict(type=float, default=0.5)),
    ('--dist_contrast', dict(type=float, default=0.5)),
    ('--dist_saturation', dict(type=float, default=0.5)),
    ('--dist_hue', dict(type=float, default=0.5)),
    ('--distortion_prob', dict(type=float, default=0.5)),
    # model
    ('--backbone', dict(type=str, default='swin_transformer')),
    ('--pretrained_file', dict(type=str,
                               default='./pretrained_weights/swin_base_patch4_window7_224_22k.pth')),
    ('--position_embedding', dict(type=str, default='sine')),
    ('--tfm_hidden_dim', dict(type=int, default=512)),
    ('--tfm_dropout', dict(type=float, default=0.1)),
    ('--tfm_nheads', dict(type=int, default=8)),
    ('--tfm_dim_feedforward', dict(type=int, default=2048)),
    ('--tfm_dec_layers', dict(type=int, default=4)),
    ('--tfm_pre_norm', dict(action='store_true')),
    # training switches that also shape inference
    ('--use_char_window_prompt', dict(action='store_true')),
    ('--global_prob', dict(type=float, default=0.4)),
    ('--use_fpn', dict(action='store_true')),
    ('--train_vie', dict(action='store_true')),
    ('--continue_train', dict(action='store_true')),
    ('--vie_categories', dict(type=int, default=0)),
    ('--lr', dict(type=float, default=0.0005)),
    ('--end_lr', dict(type=float, default=0)),
    ('--decay_power', dict(type=float, default=1)),
    ('--warmup_steps', dict(type=int, default=10000)),
    ('--max_steps', dict(type=int, default=400000)),
    ('--lr_backbone_ratio', dict(type=float, default=0.1)),
    ('--weight_decay', dict(type=float, default=1e-4)),
    ('--batch_size', dict(type=int, default=1)),
    ('--num_workers', dict(type=int, default=8)),
    ('--pt_eos_loss_coef', dict(type=float, default=0.01)),
    ('--pt_loss_weight', dict(type=float, default=1)),
    ('--poly_loss_weight', dict(type=float, default=1)),
    ('--rec_loss_weight', dict(type=float, default=1)),
    ('--epochs', dict(type=int, default=1000000000000000)),
    ('--seed', dict(type=int, default=42)),
    ('--eval', dict(action='store_true')),
    ('--resume', dict(type=str, default='')),
    ('--output_folder', dict(type=str)),
    ('--print_freq', dict(type=int, default=10)),
    ('--checkpoint_freq', dict(type=int, default=1)),
    ('--max_norm', dict(type=float, default=0.1)),
    # inference
    ('--visualize', dict(action='store_true')),
    ('--infer_vie', dict(action='store_true')),
    # distributed
    ('--local_rank', dict(type=int, default=0)),
]

# Engine-only flags (not in the reference): precision of the HIP path.
_ENGINE_FLAGS = [
    ('--engine_dtype', dict(type=str, default='bf16', choices=['bf16', 'fp32', 'bf16x3'])),
    # checkpoints (--resume, --pretrained_file) are read with torch.load(weights_only=True); files that hold other pickled
    # objects (the official Swin ImageNet files carry a 'config') need this flag -- full unpickling can execute code
    ('--allow_unsafe_pickle', dict(action='store_true')),
]


def finalize_args(args):
    """Derive the vocabulary indices exactly as reference parser.py:91-103."""
    n_char = len(args.chars) + 1  # +1 'unknown'
    args.recog_pad_index = args.num_bins + n_char
    args.pt_eos_index = args.recog_pad_index + 1
    args.poly_eos_index = args.recog_pad_index + 2
    args.rec_eos_index = args.recog_pad_index + 3
    args.pt_sos_index = args.recog_pad_index + 4
    args.poly_sos_index = args.recog_pad_index + 5
    args.rec_sos_index = args.recog_pad_index + 6
    args.padding_index = args.recog_pad_index + 7
    args.num_classes = args.padding_index + 1 + args.vie_categories
    if not hasattr(args, 'distributed'):
        args.distributed = False
    return args


class DefaultParser(object):
    """Same public surface as the reference class: add_argument() / parse_args()."""

    def __init__(self):
        self.parser = argparse.ArgumentParser()
        for flag, kw in _FLAGS + _ENGINE_FLAGS:
            self.parser.add_argument(flag, **kw)

    def add_argument(self, *a, **kw):
        self.parser.add_argument(*a, **kw)

    def parse_args(self, argv=None):
        return finalize_args(self.parser.parse_args(argv))


def make_args(**overrides):
    """Programmatic construction (tests / bench): defaults + overrides, then derive indices."""
    args = DefaultParser().parser.parse_args([])
    for k, v in overrides.items():
        setattr(args, k, v)
    return finalize_args(args)
