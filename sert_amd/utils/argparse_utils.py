"""argparse type validators (cvangysel.argparse_utils call sites:
bin/train.py:32-59, bin/query.py:30-43)."""
import argparse
import os


def existing_file_path(value):
    value = str(value)
    if not os.path.isfile(value):
        raise argparse.ArgumentTypeError('File "{0}" does not exist.'.format(value))
    return value


def nonexisting_file_path(value):
    value = str(value)
    if os.path.exists(value):
        raise argparse.ArgumentTypeError('File "{0}" already exists.'.format(value))
    return value


def positive_int(value):
    try:
        ivalue = int(value)
    except ValueError:
        raise argparse.ArgumentTypeError('"{0}" is not an integer.'.format(value))
    if ivalue <= 0:
        raise argparse.ArgumentTypeError('"{0}" is not a positive integer.'.format(value))
    return ivalue


def ratio(value):
    try:
        fvalue = float(value)
    except ValueError:
        raise argparse.ArgumentTypeError('"{0}" is not a number.'.format(value))
    if not 0.0 <= fvalue <= 1.0:
        raise argparse.ArgumentTypeError('"{0}" is not in [0, 1].'.format(value))
    return fvalue
