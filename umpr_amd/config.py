"""Configuration / command-line surface of the reference, name for name (reference config.py:7-64).

Every option of the reference exists with the same name, default and parsing rule: ``--<name> <value>``, where values of
options whose default is a bool / int / float / list go through ``ast.literal_eval`` (so ``--review_net_only True``,
``--views "['food','inside']"``), strings are taken as they are, and the two consistency checks of the reference are
kept.  Built differently from the reference: one table of fields drives the parser, the defaults and ``str()``.
Additions: ``--views food,inside`` (a plain comma list, which literal_eval rejects) is accepted besides the list literal;
``Config(argv=...)`` parses an explicit argument list (tests, tools); ``Config.extend({...})`` registers options the
reference does not have (main.py: synthetic data, loader workers, resume).
"""
import argparse
import ast

import torch

# (name, default, what it controls)                                                    default's line in config.py
_FIELDS = [
    ("device", torch.device("cuda:0"), "device the model lives on"),                                       # 8
    ("multi_gpu", True, "reference: nn.DataParallel; here one process per GPU + RCCL (umpr_amd/parallel.py)"),  # 10
    ("train_epochs", 20, "epochs"),                                                                        # 11
    ("batch_size", 64, "samples per optimiser step (per process)"),                                        # 12
    ("learning_rate", 1e-6, "Adam learning rate"),                                                         # 13
    ("l2_regularization", 1e-3, "coupled L2 on parameters whose name lacks 'bias'"),                       # 14
    ("lr_decay", 0.99, "ExponentialLR factor per epoch"),                                                  # 15
    ("word2vec_file", "embedding/glove.6B.50d.txt", "GloVe text file"),                                    # 17
    ("data_dir", "data/music", "train.csv / valid.csv / test.csv / photos.json / photos/"),                # 18
    ("log_path", "", "log file"),                                                                          # 19
    ("model_path", "", "checkpoint path"),                                                                 # 20
    ("test_only", False, "evaluate model_path on test.csv only"),                                          # 22
    ("review_net_only", False, "UMPR-R: no control net, no visual net"),                                   # 23
    ("review_level", "sentence", "'sentence' or 'review'"),                                                # 25
    ("max_sent_count", 20, "sentences kept per user / item"),                                              # 26
    ("min_sent_count", 5, "samples with fewer sentences are dropped"),                                     # 27
    ("max_ui_sent_count", 5, "sentences kept of the user's review of the item"),                           # 28
    ("max_sent_length", 20, "tokens kept per sentence"),                                                   # 29
    ("views", ["unknown"], "photo views: 1 for Amazon, ['food','inside','outside','drink'] for Yelp"),     # 30
    ("photo_count", 1, "photos per view"),                                                                 # 32
    ("gru_size", 64, "GRU hidden size (u)"),                                                               # 34
    ("self_atte_size", 64, "S-Net attention size (us)"),                                                   # 35
    ("kernel_count", 120, "C-Net Conv1d filters"),                                                         # 36
    ("kernel_size", 3, "C-Net Conv1d width"),                                                              # 37
    ("threshold", 0.35, "C-Net view-probability threshold"),                                               # 38
    ("loss_v_rate", 0.1, "weight of loss_v"),                                                              # 39
]


def _literal_or_csv(text):
    try:
        return ast.literal_eval(text)
    except (ValueError, SyntaxError):
        return [t.strip() for t in text.strip("[]").split(",") if t.strip()]


def _converter(default):
    if isinstance(default, list):
        return _literal_or_csv
    if isinstance(default, (bool, int, float)):
        return ast.literal_eval
    return type(default)


class Config:
    _extra = []   # (name, default, help) registered with extend()

    @classmethod
    def extend(cls, options):
        known = {n for n, _, _ in _FIELDS} | {n for n, _, _ in cls._extra}
        cls._extra = cls._extra + [(k, v, "") for k, v in options.items() if k not in known]

    @classmethod
    def fields(cls):
        return _FIELDS + cls._extra

    def __init__(self, argv=None):
        parser = argparse.ArgumentParser(description="UMPR on MI355X - the reference's options")
        for name, default, text in self.fields():
            parser.add_argument("--" + name, dest=name, type=_converter(default), default=default, help=text)
        for name, value in vars(parser.parse_args(argv)).items():
            setattr(self, name, value)
        if self.test_only and self.model_path == "":
            raise AssertionError("You must give model_path on testing!")
        if self.review_level not in ("sentence", "review"):
            raise AssertionError('"review_level" must be equal to "sentence" or "review"!')

    def __str__(self):
        names = sorted(set(n for n, _, _ in self.fields()) | {k for k in vars(self) if not k.startswith("_")})
        return "".join(f"{n} = {getattr(self, n)}\n" for n in names)
