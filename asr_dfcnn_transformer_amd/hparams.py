"""Hyper-parameter classes with the reference's names and defaults (util/hparams.py:5-91),
in a side-effect-free form: the reference parses sys.argv at class-definition time; here
``X().args`` / ``X.args`` give the same namespaces from defaults, and ``X.parse(argv)``
parses explicitly.  (``type=bool`` flags of the reference are always-True when given; they
are plain booleans here.)"""
import argparse


class _Hp:
    _spec = ()

    @classmethod
    def parser(cls):
        p = argparse.ArgumentParser()
        for name, default, typ in cls._spec:
            if typ is bool:
                p.add_argument('--' + name, default=default, type=lambda s: str(s).lower() not in ('0', 'false', ''))
            else:
                p.add_argument('--' + name, default=default, type=typ)
        return p

    @classmethod
    def parse(cls, argv=None):
        return cls.parser().parse_args([] if argv is None else argv)

    def __init__(self, argv=None):
        self.args = self.parse(argv)


class _ArgsDescriptor:
    def __get__(self, obj, owner):
        if obj is not None:
            return obj.__dict__['args']
        return owner.parse([])


class AmLmHparams(_Hp):
    """util/hparams.py:5-34"""
    _spec = (('am_lr', 0.0007, float), ('lm_lr', 0.00005, float), ('dacay_step', 5000, int),
             ('min_learning_rate', 1e-6, float), ('gpu_nums', 1, int), ('is_training', True, bool),
             ('am_batch_size', 16, int), ('lm_batch_size', 64, int), ('epochs', 100, int),
             ('feature_dim', 200, int), ('feature_max_length', 1600, int),
             ('num_heads', 8, int), ('num_blocks', 12, int), ('position_max_length', 100, int),
             ('max_length', 500, int), ('hidden_units', 512, int), ('dropout_rate', 0.5, float),
             ('count', 5000, int), ('crf_batch_size', 8, int), ('embedding_size', 300, int),
             ('hidden_dim', 300, int), ('keep_prob', 0.5, float), ('clip_grad', 5.0, float))


class AmDataHparams(_Hp):
    """util/hparams.py:37-53"""
    _spec = (('thchs30', True, bool), ('aishell', True, bool), ('prime', True, bool), ('stcmd', True, bool),
             ('aidatatang', False, bool), ('aidatatang_1505', False, bool), ('noise', False, bool),
             ('pinyin_dict', 'mixdict.txt', str), ('hanzi_dict', 'hanzi.txt', str),
             ('lfr_m', 4, int), ('lfr_n', 3, int))


class LmDataHparams(AmDataHparams):
    """util/hparams.py:56-72"""


class TransDataHparams(AmDataHparams):
    """util/hparams.py:75-91: as AmDataHparams except that prime and stcmd default to False (:79-80)."""
    _spec = tuple((n, (False if n in ('prime', 'stcmd') else d), t) for n, d, t in AmDataHparams._spec)


for _c in (AmLmHparams, AmDataHparams, LmDataHparams, TransDataHparams):
    _c.args = _ArgsDescriptor()
