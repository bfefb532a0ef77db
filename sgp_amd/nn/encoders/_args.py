"""CLI-flag plumbing shared by the encoders.  The reference registers flags on a
``test_tube.HyperOptArgumentParser`` through ``opt_list`` (e.g.
``lib/nn/encoders/sgp_encoder.py:53-80``); that package is optional here, so the flags
degrade to plain ``argparse`` when the parser has no ``opt_list``."""


def str_to_bool(value):
    # tsl/utils/parser_utils.py:8-15
    if isinstance(value, bool):
        return value
    if value.lower() in {'false', 'f', '0', 'no', 'n', 'off'}:
        return False
    elif value.lower() in {'true', 't', '1', 'yes', 'y', 'on'}:
        return True
    raise ValueError(f'{value} is not a valid boolean value')


def opt_list(parser, *args, tunable=False, options=None, **kwargs):
    if hasattr(parser, 'opt_list'):
        parser.opt_list(*args, tunable=tunable, options=options, **kwargs)
    else:
        parser.add_argument(*args, **kwargs)


def add_reservoir_args(parser):
    opt_list(parser, '--reservoir-size', type=int, default=32, tunable=True,
             options=[16, 32, 64, 128, 256])
    opt_list(parser, '--reservoir-layers', type=int, default=1, tunable=True, options=[1, 2, 3])
    opt_list(parser, '--spectral-radius', type=float, default=0.9, tunable=True,
             options=[0.7, 0.8, 0.9])
    opt_list(parser, '--leaking-rate', type=float, default=0.9, tunable=True,
             options=[0.7, 0.8, 0.9])
    opt_list(parser, '--density', type=float, default=0.7, tunable=True, options=[0.7, 0.8, 0.9])
    opt_list(parser, '--input-scaling', type=float, default=1., tunable=True,
             options=[1., 1.5, 2.])
    opt_list(parser, '--alpha-decay', type=str_to_bool, nargs='?', const=True, default=False)
    parser.add_argument('--reservoir-activation', type=str, default='tanh')


def add_spatial_args(parser):
    opt_list(parser, '--receptive-field', type=int, default=1, tunable=True, options=[1, 2, 3])
    opt_list(parser, '--bidirectional', type=str_to_bool, nargs='?', const=True, default=False)
    opt_list(parser, '--undirected', type=str_to_bool, nargs='?', const=True, default=False)
    opt_list(parser, '--add-self-loops', type=str_to_bool, nargs='?', const=True, default=False)
    opt_list(parser, '--global-attr', type=str_to_bool, nargs='?', const=True, default=False)
