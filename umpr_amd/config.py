"""Command-line / configuration surface of the reference (config.py:7-64), kept name for name.

Every class attribute is a default and becomes ``--<name> <value>``; bool/int/float/list values are parsed with
``ast.literal_eval`` exactly like the reference (config.py:45-52).  Additions: ``--views food,inside`` (a plain
comma list, which literal_eval rejects) is accepted besides the list literal, and ``argv`` can be injected.
"""
import argparse
import ast
import inspect

import torch


def _literal_or_csv(text):
    try:
        return ast.literal_eval(text)
    except (ValueError, SyntaxError):
        return [t.strip() for t in text.strip("[]").split(",") if t.strip()]


class Config:
    device = torch.device("cuda:0")
    multi_gpu = True  # reference: nn.DataParallel; here: one process per GPU + RCCL all-reduce (umpr_amd/parallel.py)
    train_epochs = 20
    batch_size = 64
    learning_rate = 1e-6
    l2_regularization = 1e-3
    lr_decay = 0.99

    word2vec_file = 'embedding/glove.6B.50d.txt'
    data_dir = 'data/music'
    log_path = ''
    model_path = ''

    test_only = False
    review_net_only = False

    review_level = 'sentence'
    max_sent_count = 20
    min_sent_count = 5
    max_ui_sent_count = 5
    max_sent_length = 20
    views = ['unknown']
    photo_count = 1

    gru_size = 64
    self_atte_size = 64
    kernel_count = 120
    kernel_size = 3
    threshold = 0.35
    loss_v_rate = 0.1

    def __init__(self, argv=None):
        attributes = inspect.getmembers(self, lambda a: not inspect.isfunction(a) and not inspect.ismethod(a))
        attributes = [a for a in attributes if not a[0].startswith('__')]
        parser = argparse.ArgumentParser()
        for key, val in attributes:
            receive_type = type(val)
            if receive_type is list:
                receive_type = _literal_or_csv
            elif receive_type in (bool, int, float):
                receive_type = ast.literal_eval
            parser.add_argument('--' + key, dest=key, type=receive_type, default=val)
        for key, val in parser.parse_args(argv).__dict__.items():
            setattr(self, key, val)
        if self.test_only:
            assert self.model_path != '', 'You must give model_path on testing!'
        assert self.review_level in ['sentence', 'review'], '"review_level" must be equal to "sentence" or "review"!'

    def __str__(self):
        attributes = inspect.getmembers(self, lambda a: not inspect.isfunction(a) and not inspect.ismethod(a))
        attributes = [a for a in attributes if not a[0].startswith('__')]
        return ''.join('{} = {}\n'.format(k, v) for k, v in attributes)
